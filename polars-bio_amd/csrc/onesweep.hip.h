// onesweep.hip.h -- index build, round 2: one-sweep LSD radix sort of 16-byte build records + single-pass scans.
//
// The round-1 build was launch- and gather-bound, not bandwidth-bound: 5 passes x (histogram, 3-launch scan, scatter)
// over (key, row) pairs, then random gathers of start / end / contig by row (k_index_finalize moved 1.3 GB for 5 M
// rows) and three more 3-launch scans: ~45 launches, ~1.0 ms for 5 M rows (25 % of a config-3 step).  Here:
//
//   k_ix_minmax   min / max of the starts (keys are made relative to the minimum, so 24 contigs x 28-bit coordinates sort in
//                 three 11-bit passes; a 4th launch exits at once), inverted-row flag
//   k_os_hist     per-(digit, chunk) histogram of one pass
//   k_scan_lb     single-launch look-back scan: the histogram of a pass -> write offsets; the table's max-scan
//   k_os_scatter  one digit pass: big chunks walked in sub-tiles with the digits' running write offsets in LDS, stable
//                 match-any ranking, records staged in LDS in sorted order and copied out as runs; the records carry
//                 {start, end, row, contig}, so nothing is gathered afterwards
//   k_ix_final    sorted records -> b_start / b_row / b_contig / (end, prefix max) with a look-back max-scan over the
//                 (contig, end) composites, segment offsets
//
// Inter-workgroup protocol (MI355X_MICROARCH.md, "Workgroup dispatch ... visibility"): a tile's status is ONE naturally
// aligned word {flag, value} written and read with relaxed agent-scope atomics (sc1: bypasses the non-coherent L1 / L2
// paths); no separate flag, so no ordering between two stores is needed.  Tiles are handed out by an atomic ticket, so
// every predecessor of a tile is owned by a workgroup that is already running: the look-back cannot deadlock whatever the
// dispatch order.
#pragma once
#include "index_view.hip.h"
#include "slice.hip.h"

namespace ivj {

constexpr int OS_THREADS = 1024;
static_assert(OS_THREADS == SL_THREADS, "the workgroup scans of slice.hip.h are shared");
constexpr int OS_WAVES = OS_THREADS / kWave;
constexpr int OS_ITEMS = 4;
constexpr int OS_TILE = OS_THREADS * OS_ITEMS;
constexpr int OS_BITS = 11;
constexpr int OS_RADIX = 1 << OS_BITS;
constexpr int OS_MAX_PASSES = 6;                       // 32 coordinate bits + up to 30 contig bits
constexpr int OS_LDS_CONTIGS = 1023;                   // per-contig statistics are privatised in LDS up to this many contigs

constexpr uint32_t OS_FLAG_AGG = 1u << 30, OS_FLAG_PRE = 2u << 30, OS_VAL_MASK = (1u << 30) - 1u;

struct OsMeta {                                        // device-resident, zero-initialised before every build
    uint32_t umax;                                     // max of flip(start)
    uint32_t inv_umin;                                 // max of ~flip(start)  (=> umin = ~inv_umin; both grow from 0)
    uint32_t inverted;                                 // some row of the dictionary has start > end
    uint32_t pad;
    uint32_t ticket[OS_MAX_PASSES + 2];                // tile tickets: one per pass, [OS_MAX_PASSES] final, [+1] table scan
};

__device__ __forceinline__ uint32_t os_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void os_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long os_ld64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void os_st64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__host__ __device__ __forceinline__ int os_bits_for(uint32_t v) { int b = 0; while (b < 32 && (v >> b) != 0) ++b; return b == 0 ? 1 : b; }

// key geometry shared by the histogram and the pass kernels (uniform): key = (contig' << sbits) | (flip(start) - umin)
struct OsKey {
    uint32_t umin;
    int sbits, total;
};
__device__ __forceinline__ OsKey os_key_geom(const OsMeta* meta, int cbits) {
    OsKey k;
    k.umin = ~meta->inv_umin;
    k.sbits = os_bits_for(meta->umax - k.umin);
    k.total = k.sbits + cbits;
    return k;
}
__device__ __forceinline__ unsigned long long os_key(const OsKey& k, uint32_t c, int32_t start) {
    return ((unsigned long long)c << k.sbits) | (unsigned long long)(flip(start) - k.umin);
}

// ---- statistics: min / max of the starts (the sort keys are made relative to the minimum), inverted-row flag ----------
__global__ __launch_bounds__(OS_THREADS) void k_ix_minmax(const int32_t* __restrict__ start, const int32_t* __restrict__ end,
                                                         const int32_t* __restrict__ contig, int64_t n, int32_t n_contigs, OsMeta* __restrict__ meta) {
    uint32_t mx = 0, imn = 0, inv = 0;
    for (int64_t i = (int64_t)blockIdx.x * OS_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * OS_THREADS) {
        const int32_t s = start[i];
        const uint32_t u = flip(s);
        mx = u > mx ? u : mx;
        imn = ~u > imn ? ~u : imn;
        if (s > end[i] && (uint32_t)contig[i] < (uint32_t)n_contigs) inv = 1;
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        const uint32_t a = __shfl_xor(mx, d, kWave), b = __shfl_xor(imn, d, kWave), c = __shfl_xor(inv, d, kWave);
        mx = a > mx ? a : mx; imn = b > imn ? b : imn; inv |= c;
    }
    // same-address atomics complete one per ~12 ns on this part: one attempt per workgroup, and only when it can change the value
    __shared__ uint32_t l_mx[OS_WAVES], l_imn[OS_WAVES], l_inv[OS_WAVES];
    const int w = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) { l_mx[w] = mx; l_imn[w] = imn; l_inv[w] = inv; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 1; k < OS_WAVES; ++k) { mx = l_mx[k] > mx ? l_mx[k] : mx; imn = l_imn[k] > imn ? l_imn[k] : imn; inv |= l_inv[k]; }
        if (mx > os_ld(&meta->umax)) atomicMax(&meta->umax, mx);
        if (imn > os_ld(&meta->inv_umin)) atomicMax(&meta->inv_umin, imn);
        if (inv && !os_ld(&meta->inverted)) atomicOr(&meta->inverted, 1u);
    }
}

// ---- one digit pass -------------------------------------------------------------------------------------------------------
// Round-2 note: a decoupled look-back over 2048 digits per tile was measured first (0.16 ms per pass for 5 M records: every
// thread walks back over the tiles that run concurrently, 8 KB of status words per step).  The passes therefore use BIG tiles
// with running offsets instead: workgroup g owns the contiguous chunk [g * chunk, (g + 1) * chunk) of the pass's input and
// walks it in sub-tiles of OS_TILE records, keeping the global write offset of every digit in LDS.  Its starting offsets come
// from the exclusive scan of the per-(digit, chunk) histogram `off` (digit-major, one look-back scan launch per pass).
// (Feeding the next pass's histogram from this pass's scatter -- one global atomic per record -- was measured as well: 5 M
// scattered atomics cost more than the 20 us histogram launch they save.)
struct OsPassLds { int wcnt, rec, d, base, lstart, wsum, total; };
__host__ __device__ inline OsPassLds os_pass_lds() {
    OsPassLds L;
    int o = 0;
    L.rec = o; o += 16 * OS_TILE;
    L.wcnt = o; o += 2 * OS_RADIX * OS_WAVES;
    L.d = o; o += 2 * OS_TILE;
    L.base = o; o += 4 * OS_RADIX;
    L.lstart = o; o += 4 * OS_RADIX;
    L.wsum = (o + 15) & ~15; o = L.wsum + 4 * OS_WAVES;
    L.total = o;
    return L;
}

__device__ __forceinline__ int4 os_load_record(bool first, const int32_t* __restrict__ contig, const int32_t* __restrict__ start,
                                               const int32_t* __restrict__ end, const int32_t* __restrict__ row_id, const int4* __restrict__ src,
                                               int64_t i, int32_t n_contigs) {
    if (first) {
        const int32_t c0 = contig[i];
        return make_int4(start[i], end[i], row_id ? row_id[i] : (int32_t)i, (uint32_t)c0 < (uint32_t)n_contigs ? c0 : n_contigs);
    }
    return src[i];
}

// per-(digit, chunk) histogram of one pass (digit-major: hist[d * nchunks + g]); pass 0 reads the caller's columns, later
// passes the records the previous pass wrote
template <bool FIRST>
__global__ __launch_bounds__(OS_THREADS) void k_os_hist(const int32_t* __restrict__ contig, const int32_t* __restrict__ start,
                                                       const int4* __restrict__ src, int64_t n, int32_t n_contigs, int cbits, int pass,
                                                       const OsMeta* __restrict__ meta, int chunk, int nchunks, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[OS_RADIX];
    const OsKey kg = os_key_geom(meta, cbits);
    const int shift = pass * OS_BITS;
    if (shift >= kg.total) return;                                             // uniform
    for (int k = threadIdx.x; k < OS_RADIX; k += OS_THREADS) h[k] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * chunk;
    const int64_t end = base + chunk < n ? base + chunk : n;
    for (int64_t i = base + threadIdx.x; i < end; i += OS_THREADS) {
        uint32_t c; int32_t s0;
        if (FIRST) { const int32_t c0 = contig[i]; c = (uint32_t)c0 < (uint32_t)n_contigs ? (uint32_t)c0 : (uint32_t)n_contigs; s0 = start[i]; }
        else { const int4 r = src[i]; c = (uint32_t)r.w; s0 = r.x; }
        atomicAdd(&h[(int)((os_key(kg, c, s0) >> shift) & (OS_RADIX - 1))], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < OS_RADIX; k += OS_THREADS) hist[(int64_t)k * nchunks + blockIdx.x] = h[k];
}

template <bool FIRST>
__global__ __launch_bounds__(OS_THREADS) void k_os_scatter(const int32_t* __restrict__ contig, const int32_t* __restrict__ start,
                                                          const int32_t* __restrict__ end, const int32_t* __restrict__ row_id,
                                                          const int4* __restrict__ src, int4* __restrict__ dst, int64_t n, int32_t n_contigs, int cbits,
                                                          int pass, const OsMeta* __restrict__ meta, int chunk, int nchunks,
                                                          const uint32_t* __restrict__ off /* exclusive scan of this pass's histogram */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char os_lds[];
    const OsPassLds L = os_pass_lds();
    int4* l_rec = reinterpret_cast<int4*>(os_lds + L.rec);
    unsigned short* wcnt = reinterpret_cast<unsigned short*>(os_lds + L.wcnt);
    unsigned short* l_d = reinterpret_cast<unsigned short*>(os_lds + L.d);
    uint32_t* base = reinterpret_cast<uint32_t*>(os_lds + L.base);
    uint32_t* lstart = reinterpret_cast<uint32_t*>(os_lds + L.lstart);
    uint32_t* wsum = reinterpret_cast<uint32_t*>(os_lds + L.wsum);

    const OsKey kg = os_key_geom(meta, cbits);
    const int shift = pass * OS_BITS;
    if (shift >= kg.total) return;                                             // uniform: every key bit is sorted already
    const int tid = threadIdx.x, w = tid / kWave, lane = tid & (kWave - 1);
    for (int k = tid; k < OS_RADIX; k += OS_THREADS) base[k] = off[(int64_t)k * nchunks + blockIdx.x];
    for (int k = tid; k < OS_RADIX * OS_WAVES / 2; k += OS_THREADS) reinterpret_cast<uint32_t*>(wcnt)[k] = 0;
    __syncthreads();
    const int64_t cbase = (int64_t)blockIdx.x * chunk;
    const int64_t cend = cbase + chunk < n ? cbase + chunk : n;
    const uint64_t lt = lanemask_lt();
    unsigned short* my = wcnt + w * OS_RADIX;
    // wavefront w owns elements [w * 256, (w + 1) * 256) of the sub-tile: item j of lane l = w * 256 + j * 64 + l
    const int el0 = w * (OS_ITEMS * kWave) + lane;
    // the next sub-tile's records are requested before this one is ranked: a pass walks ~5 sub-tiles per workgroup with seven
    // barriers each and nothing else resident on the CU, so an exposed load is a stalled CU
    int4 nxt[OS_ITEMS];
    auto load_tile = [&](int64_t tb) {
        const int tn = (int)((cend - tb) < (int64_t)OS_TILE ? (cend - tb) : (int64_t)OS_TILE);
#pragma unroll
        for (int j = 0; j < OS_ITEMS; ++j) {
            const int il = el0 + j * kWave;
            nxt[j] = il < tn ? os_load_record(FIRST, contig, start, end, row_id, src, tb + il, n_contigs) : make_int4(0, 0, 0, 0);
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
            d[j] = (uint32_t)((os_key(kg, (uint32_t)r[j].w, r[j].x) >> shift) & (OS_RADIX - 1));
        }
        if (tbase + OS_TILE < cend) load_tile(tbase + OS_TILE);
        // stable rank inside (wavefront, digit): match-any ballots against the wavefront's private counter row
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
        // thread t owns digits 2t, 2t+1: exclusive prefix over the wavefronts, sub-tile totals, sub-tile-local starts
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
        base[2 * tid] += x0;                                                   // this thread's two digits: nobody else touches them
        base[2 * tid + 1] += x1;
        __syncthreads();
    }
}

// ---- sorted records -> index arrays --------------------------------------------------------------------------------------
// status64[tile] = flag << 62 | composite: AGG = max (contig, end) composite of the tile, PRE = max over tiles 0..tile.
// The composite carries the contig in its high word, so the prefix max never leaks across a contig boundary.
__global__ __launch_bounds__(OS_THREADS) void k_ix_final(const int4* __restrict__ recA, const int4* __restrict__ recB, int64_t n, int32_t n_contigs,
                                                        int cbits, OsMeta* __restrict__ meta, unsigned long long* __restrict__ status64,
                                                        int32_t* __restrict__ b_start, int2* __restrict__ ep, int32_t* __restrict__ b_row,
                                                        int32_t* __restrict__ b_contig, int32_t* __restrict__ seg, int32_t* __restrict__ flags) {
    __shared__ unsigned long long wmax[OS_WAVES];
    __shared__ unsigned long long s_carry;
    __shared__ int l_tile;
    const OsKey kg = os_key_geom(meta, cbits);
    const int passes = (kg.total + OS_BITS - 1) / OS_BITS;
    const int4* __restrict__ rec = ((passes - 1) & 1) ? recB : recA;           // pass p writes buffer p & 1 (A, B, A, ...)
    const int tid = threadIdx.x, w = tid / kWave, lane = tid & (kWave - 1);
    if (tid == 0) l_tile = (int)atomicAdd(&meta->ticket[OS_MAX_PASSES], 1u);
    __syncthreads();
    const int tile = l_tile;
    const int64_t i0 = (int64_t)tile * OS_TILE + (int64_t)tid * OS_ITEMS;      // four consecutive rows per thread
    int4 r[OS_ITEMS];
    unsigned long long comp[OS_ITEMS];
    unsigned long long run = 0;
#pragma unroll
    for (int j = 0; j < OS_ITEMS; ++j) {
        r[j] = i0 + j < n ? rec[i0 + j] : make_int4(0, 0, 0, 0);
        const unsigned long long c = i0 + j < n ? (((unsigned long long)(uint32_t)r[j].w << 32) | (unsigned long long)flip(r[j].y)) : 0ull;
        run = c > run ? c : run;
        comp[j] = run;                                                         // inclusive max inside the thread
    }
    // workgroup inclusive max-scan of the thread maxima
    unsigned long long inc = run;
#pragma unroll
    for (int dd = 1; dd < kWave; dd <<= 1) {
        const unsigned long long o = __shfl_up(inc, dd, kWave);
        if (lane >= dd) inc = o > inc ? o : inc;
    }
    if (lane == kWave - 1) wmax[w] = inc;
    __syncthreads();
    unsigned long long wpre = 0, tmax = 0;
#pragma unroll
    for (int k = 0; k < OS_WAVES; ++k) { const unsigned long long x = wmax[k]; if (k < w) wpre = x > wpre ? x : wpre; tmax = x > tmax ? x : tmax; }
    unsigned long long excl = __shfl_up(inc, 1, kWave);
    if (lane == 0) excl = 0;
    excl = wpre > excl ? wpre : excl;                                          // max over the tile's rows before this thread
    // look-back over the earlier tiles (one thread), broadcast through LDS
    if (tid == 0) {
        const unsigned long long VAL = (1ull << 62) - 1ull;
        unsigned long long carry = 0;
        if (tile == 0) os_st64(status64, (2ull << 62) | tmax);
        else {
            os_st64(status64 + tile, (1ull << 62) | tmax);
            for (int t = tile - 1; t >= 0; --t) {
                unsigned long long v = os_ld64(status64 + t);
                while ((v >> 62) == 0) { __builtin_amdgcn_s_sleep(2); v = os_ld64(status64 + t); }
                const unsigned long long x = v & VAL;
                carry = x > carry ? x : carry;
                if ((v >> 62) == 2ull) break;
            }
            os_st64(status64 + tile, (2ull << 62) | (carry > tmax ? carry : tmax));
        }
        s_carry = carry;
    }
    __syncthreads();
    const unsigned long long before = s_carry > excl ? s_carry : excl;
    int32_t s4[OS_ITEMS], row4[OS_ITEMS], c4[OS_ITEMS];
#pragma unroll
    for (int j = 0; j < OS_ITEMS; ++j) {
        const unsigned long long pm = before > comp[j] ? before : comp[j];
        s4[j] = r[j].x; row4[j] = r[j].z; c4[j] = r[j].w;
        if (i0 + j < n) ep[i0 + j] = make_int2(r[j].y, unflip((uint32_t)pm));
    }
    if (i0 + OS_ITEMS <= n) {
        *reinterpret_cast<int4*>(b_start + i0) = make_int4(s4[0], s4[1], s4[2], s4[3]);
        *reinterpret_cast<int4*>(b_row + i0) = make_int4(row4[0], row4[1], row4[2], row4[3]);
        *reinterpret_cast<int4*>(b_contig + i0) = make_int4(c4[0], c4[1], c4[2], c4[3]);
    } else {
#pragma unroll
        for (int j = 0; j < OS_ITEMS; ++j) if (i0 + j < n) { b_start[i0 + j] = s4[j]; b_row[i0 + j] = row4[j]; b_contig[i0 + j] = c4[j]; }
    }
    // segment offsets: seg[k] = first position whose contig key is >= k, written by the row where the contig changes
    // (contig keys are 0 .. n_contigs, the last one = rows outside the dictionary)
    int32_t prev_c = -1;
    if (i0 > 0 && i0 < n) prev_c = rec[i0 - 1].w;
#pragma unroll
    for (int j = 0; j < OS_ITEMS; ++j) {
        const int64_t p = i0 + j;
        if (p >= n) break;
        for (int32_t kk = prev_c + 1; kk <= c4[j]; ++kk) seg[kk] = (int32_t)p;
        prev_c = c4[j];
        if (p == n - 1) for (int32_t kk = c4[j] + 1; kk <= n_contigs + 1; ++kk) seg[kk] = (int32_t)n;
    }
    if (tile == 0 && tid == 0) flags[0] = (int32_t)meta->inverted;
}

// ---- single-launch inclusive scan (look-back), u32 --------------------------------------------------------------------------
constexpr int LB_ITEMS = 8;
constexpr int LB_TILE = OS_THREADS * LB_ITEMS;

template <class Op, bool EXCLUSIVE>
__global__ __launch_bounds__(OS_THREADS) void k_scan_lb_u32(uint32_t* __restrict__ data, int64_t n, uint32_t identity, uint32_t* __restrict__ ticket,
                                                           unsigned long long* __restrict__ status64) {
    __shared__ uint32_t wtot[OS_WAVES];
    __shared__ uint32_t s_carry;
    __shared__ int l_tile;
    Op op;
    const int tid = threadIdx.x, w = tid / kWave, lane = tid & (kWave - 1);
    if (tid == 0) l_tile = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int tile = l_tile;
    const int64_t i0 = (int64_t)tile * LB_TILE + (int64_t)tid * LB_ITEMS;
    uint32_t v[LB_ITEMS];
    if (i0 + LB_ITEMS <= n) {
        const uint4 a = *reinterpret_cast<const uint4*>(data + i0), b = *reinterpret_cast<const uint4*>(data + i0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < LB_ITEMS; ++j) v[j] = i0 + j < n ? data[i0 + j] : identity;
    }
    uint32_t run = identity;
#pragma unroll
    for (int j = 0; j < LB_ITEMS; ++j) { const uint32_t x = v[j]; if (EXCLUSIVE) v[j] = run; run = op(run, x); if (!EXCLUSIVE) v[j] = run; }
    uint32_t inc = run;
#pragma unroll
    for (int dd = 1; dd < kWave; dd <<= 1) {
        const uint32_t o = __shfl_up(inc, dd, kWave);
        if (lane >= dd) inc = op(o, inc);
    }
    if (lane == kWave - 1) wtot[w] = inc;
    __syncthreads();
    uint32_t wpre = identity, ttot = identity;
#pragma unroll
    for (int k = 0; k < OS_WAVES; ++k) { const uint32_t x = wtot[k]; if (k < w) wpre = op(wpre, x); ttot = op(ttot, x); }
    uint32_t excl = __shfl_up(inc, 1, kWave);
    if (lane == 0) excl = identity;
    excl = op(wpre, excl);
    if (tid == 0) {
        uint32_t carry = identity;
        if (tile == 0) os_st64(status64, (2ull << 62) | ttot);
        else {
            os_st64(status64 + tile, (1ull << 62) | ttot);
            for (int t = tile - 1; t >= 0; --t) {
                unsigned long long x = os_ld64(status64 + t);
                while ((x >> 62) == 0) { __builtin_amdgcn_s_sleep(2); x = os_ld64(status64 + t); }
                carry = op((uint32_t)x, carry);
                if ((x >> 62) == 2ull) break;
            }
            os_st64(status64 + tile, (2ull << 62) | op(carry, ttot));
        }
        s_carry = carry;
    }
    __syncthreads();
    const uint32_t before = op(s_carry, excl);
#pragma unroll
    for (int j = 0; j < LB_ITEMS; ++j) v[j] = op(before, v[j]);
    if (i0 + LB_ITEMS <= n) {
        *reinterpret_cast<uint4*>(data + i0) = make_uint4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<uint4*>(data + i0 + 4) = make_uint4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
        for (int j = 0; j < LB_ITEMS; ++j) if (i0 + j < n) data[i0 + j] = v[j];
    }
}

// ---- single-launch sum scan (look-back), uint32 / int64, in -> out (may alias), optional grand total -----------------------------
// What the count -> fill pair, the sort-scan family and the cluster ids used three launches for (reduce, partials, apply: the input
// read twice).  Status word = flag << 62 | running sum (sums below 2^62); tiles handed out by ticket, as above.
// ITEMS per thread: 8 (tiles of 8192 elements) or 32 for long inputs -- a tile's fixed cost (ticket, three barriers, the look-back) is
// ~ 10 us, so 100 M elements in 8192-element tiles cost more in tiles than in bytes (0.57 -> 0.3 ms)
template <class T, bool INCLUSIVE, int ITEMS>
__global__ __launch_bounds__(OS_THREADS) void k_scan_lb_sum(const T* __restrict__ in, T* __restrict__ out, int64_t n, uint32_t* __restrict__ ticket,
                                                           unsigned long long* __restrict__ status64, T* __restrict__ total_out) {
    __shared__ unsigned long long wtot[OS_WAVES];
    __shared__ unsigned long long s_carry;
    __shared__ int l_tile;
    const int tid = threadIdx.x, w = tid / kWave, lane = tid & (kWave - 1);
    if (tid == 0) l_tile = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int tile = l_tile;
    constexpr int TILE = OS_THREADS * ITEMS;
    const int64_t i0 = (int64_t)tile * TILE + (int64_t)tid * ITEMS;
    unsigned long long v[ITEMS];
    constexpr int PER16 = 16 / (int)sizeof(T);                                  // elements per 16-byte load
    const bool whole = i0 + ITEMS <= n && (reinterpret_cast<uintptr_t>(in) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
    if (whole) {
#pragma unroll
        for (int q = 0; q < ITEMS / PER16; ++q) {
            const uint4 x = *reinterpret_cast<const uint4*>(in + i0 + q * PER16);
            if constexpr (sizeof(T) == 4) { v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w; }
            else { v[2 * q] = (unsigned long long)x.x | ((unsigned long long)x.y << 32); v[2 * q + 1] = (unsigned long long)x.z | ((unsigned long long)x.w << 32); }
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) v[j] = i0 + j < n ? (unsigned long long)in[i0 + j] : 0ull;
    }
    unsigned long long run = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { const unsigned long long x = v[j]; if (!INCLUSIVE) v[j] = run; run += x; if (INCLUSIVE) v[j] = run; }
    unsigned long long inc = run;
#pragma unroll
    for (int dd = 1; dd < kWave; dd <<= 1) {
        const unsigned long long o = __shfl_up(inc, dd, kWave);
        if (lane >= dd) inc += o;
    }
    if (lane == kWave - 1) wtot[w] = inc;
    __syncthreads();
    unsigned long long wpre = 0, ttot = 0;
#pragma unroll
    for (int k = 0; k < OS_WAVES; ++k) { const unsigned long long x = wtot[k]; if (k < w) wpre += x; ttot += x; }
    unsigned long long excl = __shfl_up(inc, 1, kWave);
    if (lane == 0) excl = 0;
    excl += wpre;
    if (tid == 0) {
        constexpr unsigned long long VAL = (1ull << 62) - 1ull;
        unsigned long long carry = 0;
        if (tile == 0) os_st64(status64, (2ull << 62) | (ttot & VAL));
        else {
            os_st64(status64 + tile, (1ull << 62) | (ttot & VAL));
            for (int t = tile - 1; t >= 0; --t) {
                unsigned long long x = os_ld64(status64 + t);
                while ((x >> 62) == 0) { __builtin_amdgcn_s_sleep(2); x = os_ld64(status64 + t); }
                carry += x & VAL;
                if ((x >> 62) == 2ull) break;
            }
            os_st64(status64 + tile, (2ull << 62) | ((carry + ttot) & VAL));
        }
        s_carry = carry;
        if (total_out && (int64_t)(tile + 1) * TILE >= n) *total_out = (T)(carry + ttot);
    }
    __syncthreads();
    const unsigned long long before = s_carry + excl;
    if (whole) {
#pragma unroll
        for (int q = 0; q < ITEMS / PER16; ++q) {
            uint4 x;
            if constexpr (sizeof(T) == 4) {
                x = make_uint4((uint32_t)(before + v[4 * q]), (uint32_t)(before + v[4 * q + 1]), (uint32_t)(before + v[4 * q + 2]), (uint32_t)(before + v[4 * q + 3]));
            } else {
                const unsigned long long a = before + v[2 * q], b = before + v[2 * q + 1];
                x = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
            }
            *reinterpret_cast<uint4*>(out + i0 + q * PER16) = x;
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) if (i0 + j < n) out[i0 + j] = (T)(before + v[j]);
    }
}

// ---- end order: the same sort over (contig, end) with the position as the record's row --------------------------------------
__global__ void k_end_column(const int2* __restrict__ ep, int64_t n, int32_t* __restrict__ ends) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ends[i] = ep[i].x;
}
// sorted records {end, -, position in the start order, contig} -> e_end / e_pos
__global__ void k_end_unpack(const int4* __restrict__ recA, const int4* __restrict__ recB, int64_t n, int cbits, const OsMeta* __restrict__ meta,
                             int32_t* __restrict__ e_end, int32_t* __restrict__ e_pos) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const OsKey kg = os_key_geom(meta, cbits);
    const int passes = (kg.total + OS_BITS - 1) / OS_BITS;
    const int4 r = (((passes - 1) & 1) ? recB : recA)[i];                      // pass p writes buffer p & 1 (A, B, A, ...)
    e_end[i] = r.x; e_pos[i] = r.z;
}

}  // namespace ivj
