// flat.hip.h -- pb.overlap, load-balanced ("flat") single pass.
//
// The per-lane window scan of overlap.hip.h gives every lane a different trip count (windows of
// 1..many rows) and a chain of dependent gathers (table record -> row group -> next group -> build
// row).  Here a probe is first reduced to a LOOSE candidate range [lo', hi') with two independent
// 4-byte table reads and no search at all:
//     hi' = tab2[slot(q.end) + 1].x first position of the NEXT start bin            (>= exact hi)
//     lo' = tab2[slot(q.start)].y   first position whose prefix max reaches the lower edge of
//                                   q.start's bin                                    (<= exact lo)
// A bin is about half a build-row spacing wide, so the range holds ~1 row more than the exact
// window.  The candidates of a tile of probes are then laid out flat (exclusive scan of the range
// lengths) and EVERY LANE TESTS ONE CANDIDATE with the literal predicate
//     b.start (<) q.end  &&  q.start (<) b.end
// on one 16-byte {start, end, build row} read (no second gather for the row id): no divergence, lanes of one probe read consecutive rows, long
// (dense / nested) windows are spread over the whole workgroup by construction.  The result is exact
// for every input (inverted and zero-length rows included) because the range is a superset of the
// matches and the predicate is evaluated as written (polars_bio/range_op.py:75-84).
#pragma once
#include "index_view.hip.h"

namespace ivj {

constexpr int FLAT_THREADS = 256;
constexpr int FLAT_ITEMS = 2;
constexpr int FLAT_TILE = FLAT_THREADS * FLAT_ITEMS;   // probes per workgroup
constexpr int FLAT_MAX_CAND = 1 << 15;                  // candidates of ONE probe the flat path accepts
constexpr int FLAT_CH = 2048;                          // candidates per chunk (one 16-byte mark vector per thread)
static_assert(FLAT_CH * 2 == FLAT_THREADS * 16, "one uint4 of marks per thread");
static_assert(FLAT_TILE < 65535, "probe index + 1 must fit the 16-bit marks");

// lot (zero-filled) receives the change points of "number of bin edges the prefix max has reached";
// an inclusive max-scan over the whole table then yields
//   lot[tb + j] = first position p of the contig whose prefix max is >= ulo + (j << shift)
// (entries that inherit a smaller value from the previous contig are clamped to the segment start by
// the reader).  Slot G(b-1) gets b: no row reaches the edges above the final prefix max.
__global__ void k_lot_mark(const int2* __restrict__ ep, const int32_t* __restrict__ b_contig, int64_t n,
                           int32_t n_contigs, const int4* __restrict__ cmeta, uint32_t* __restrict__ lot) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t c = b_contig[p];
    if ((uint32_t)c >= (uint32_t)n_contigs) return;
    const int4 m0 = cmeta[2 * c], m1 = cmeta[2 * c + 1];
    const uint32_t ulo = (uint32_t)m0.z, uhi = (uint32_t)m0.w;
    const uint32_t nb = ((uhi - ulo) >> m1.x) + 1u;
    auto edges = [&](int32_t pm) -> uint32_t {
        const uint32_t u = flip(pm);
        if (u < ulo) return 0u;
        const uint32_t g = ((u - ulo) >> m1.x) + 1u;
        return g < nb ? g : nb;
    };
    const uint32_t gp = edges(ep[p].y);
    const uint32_t gprev = (p == m0.x) ? 0u : edges(ep[p - 1].y);
    if (gp > gprev) lot[(uint32_t)m1.y + gprev] = (uint32_t)p;
    if (p == m0.y - 1) lot[(uint32_t)m1.y + gp] = (uint32_t)m0.y;
}

// tab2[slot] = {bins[slot], lot[slot]}: both bounds of a read-length probe usually sit in one 64-byte line
__global__ void k_tab2(const uint32_t* __restrict__ bins, const uint32_t* __restrict__ lot, int64_t len, uint2* __restrict__ tab2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) tab2[i] = make_uint2(bins[i], lot[i]);
}

// rec4[p] = {start, end, build row, prefix max}: the one 16-byte read per candidate
__global__ void k_rec4(const int32_t* __restrict__ b_start, const int2* __restrict__ ep, const int32_t* __restrict__ b_row, int64_t n,
                       int4* __restrict__ rec4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int2 v = ep[i]; rec4[i] = make_int4(b_start[i], v.x, b_row[i], v.y); }
}

// candidate range of one probe: [lo, lo + n)
template <bool STRICT>
__device__ __forceinline__ void flat_range(const IndexView& ix, int32_t c, int32_t qs, int32_t qe, bool valid, int& lo, int& n) {
    lo = 0; n = 0;
    if (!valid || (uint32_t)c >= (uint32_t)ix.n_contigs) return;
    const int4 m0 = ix.cmeta[2 * c], m1 = ix.cmeta[2 * c + 1];
    const int a = m0.x, b = m0.y;
    const uint32_t ulo = (uint32_t)m0.z, uhi = (uint32_t)m0.w;
    // rows pass "start (<) q.end" iff ustart < te; rows pass "q.start (<) pmax" iff upmax >= tq
    const unsigned long long te = (unsigned long long)flip(qe) + (STRICT ? 0ull : 1ull);
    const unsigned long long tq = (unsigned long long)flip(qs) + (STRICT ? 1ull : 0ull);
    if (b <= a || te <= (unsigned long long)ulo) return;
    int hi = b;
    if (te <= (unsigned long long)uhi) hi = (int)ix.tab2[(uint32_t)m1.y + (((uint32_t)te - ulo) >> m1.x) + 1u].x;
    int l = a;
    if (tq > (unsigned long long)ulo) {
        const uint32_t j = tq > (unsigned long long)uhi ? ((uhi - ulo) >> m1.x) : (((uint32_t)tq - ulo) >> m1.x);
        const int t = (int)ix.tab2[(uint32_t)m1.y + j].y;
        l = t > a ? t : a;
    }
    lo = l;
    n = hi > l ? hi - l : 0;
}

// One chunk of FLAT_CH candidates starting at tile-local candidate offset c0.  Builds the
// candidate -> probe map (marks: probe index + 1 at the probe's first candidate, max-scanned), then
// wavefront w tests the candidates [w*per, (w+1)*per).  EMIT: the matches of a wavefront are staged at
// st_b / marks[w*per + rank] (marks is recycled as the probe index of the staged pair: rank <= index,
// and the lanes of a step have read their marks before any of them writes).  Returns the number of
// matches of this wavefront.
template <bool STRICT, bool EMIT>
__device__ __forceinline__ int flat_chunk(const IndexView& ix, long long c0, int nC, int per, const long long (&off)[FLAT_ITEMS],
                                          const int (&cn)[FLAT_ITEMS], const int* l_lo, const uint32_t* l_off, const int2* l_q,
                                          uint16_t* marks, int32_t* st_b, int* lds_i) {
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    const unsigned long long lt_lanes = (1ull << lane) - 1ull;
    uint4* marks4 = reinterpret_cast<uint4*>(marks);
    __syncthreads();                                       // the previous chunk is done with marks / st_b
    marks4[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < FLAT_ITEMS; ++k) {
        if (cn[k] == 0) continue;
        const long long rel = off[k] - c0;
        if (rel >= 0 && rel < (long long)nC) marks[rel] = (uint16_t)(threadIdx.x * FLAT_ITEMS + k + 1);
        else if (rel < 0 && rel + cn[k] > 0) marks[0] = (uint16_t)(threadIdx.x * FLAT_ITEMS + k + 1);
    }
    __syncthreads();
    {
        uint4 v = marks4[threadIdx.x];
        uint32_t wd[4] = {v.x, v.y, v.z, v.w};
        uint32_t tmax = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t a = wd[j] & 0xffffu, b = wd[j] >> 16;
            tmax = tmax > a ? tmax : a;
            tmax = tmax > b ? tmax : b;
        }
        int tot;
        uint32_t run = (uint32_t)block_exclusive_scan((int)tmax, MaxOp(), 0, lds_i, &tot);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t a = wd[j] & 0xffffu, b = wd[j] >> 16;
            run = run > a ? run : a;
            const uint32_t na = run;
            run = run > b ? run : b;
            wd[j] = na | (run << 16);
        }
        marks4[threadIdx.x] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
    }
    __syncthreads();
    const int wb = w * per;
    const int we = (wb + per) < nC ? (wb + per) : nC;
    const uint32_t c0lo = (uint32_t)c0;
    int cnt = 0;
    for (int i0 = wb; i0 < we; i0 += kWave) {
        const int i = i0 + lane;
        const bool valid = i < we;
        int q = valid ? (int)marks[i] - 1 : 0;
        q = q < 0 ? 0 : q;
        const int lo = l_lo[q];
        const uint32_t of = l_off[q];
        const int2 qq = l_q[q];
        const int p = lo + (int)(c0lo + (uint32_t)i - of);
        int4 v = make_int4(0, 0, 0, 0);                    // {start, end, build row, -}
        if (valid) v = ix.rec4[p];
        const bool m = valid && lt_op<STRICT>(v.x, qq.y) && lt_op<STRICT>(qq.x, v.y);
        const unsigned long long mm = __ballot(m);
        if (EMIT && m) {
            const int r = wb + cnt + (int)__popcll(mm & lt_lanes);
            st_b[r] = v.z;
            marks[r] = (uint16_t)q;
        }
        cnt += (int)__popcll(mm);
    }
    return cnt;
}

// Fused single pass with the contract of k_overlap_fused (overlap.hip.h): every tile reserves ONE
// contiguous output range with one 64-bit atomicAdd; pairs of one probe stay contiguous and ordered
// by (build.start, build row).  Tiles with more than FLAT_CH candidates (dense data) count first and
// emit in a second sweep over their chunks.
template <bool STRICT>
__global__ __launch_bounds__(FLAT_THREADS, 6) void k_overlap_flat(IndexView ix, const int32_t* __restrict__ pc,
                                                                const int32_t* __restrict__ ps,
                                                                const int32_t* __restrict__ pe,
                                                                const int32_t* __restrict__ probe_ids, int64_t n,
                                                                bool vec_ok, long long capacity,
                                                                unsigned long long* __restrict__ state,
                                                                int32_t* __restrict__ out_probe,
                                                                int32_t* __restrict__ out_build, int rank_counts) {
    __shared__ long long lds_ll[FLAT_THREADS / kWave];
    __shared__ int lds_i[FLAT_THREADS / kWave];
    __shared__ long long s_wtot[FLAT_THREADS / kWave];
    __shared__ long long s_base;
    __shared__ int l_lo[FLAT_TILE];
    __shared__ uint32_t l_off[FLAT_TILE];
    __shared__ int2 l_q[FLAT_TILE];
    __shared__ int32_t l_row[FLAT_TILE];
    __shared__ __align__(16) uint16_t marks[FLAT_CH];
    __shared__ int32_t st_b[FLAT_CH];
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    const long long ntiles = (n + FLAT_TILE - 1) / FLAT_TILE;
    const long long tile = xcd_tile64(blockIdx.x, ntiles);
    if (tile >= ntiles) return;                            // uniform
    const int64_t i0 = (int64_t)tile * FLAT_TILE + (int64_t)threadIdx.x * FLAT_ITEMS;
    int32_t c[FLAT_ITEMS], s[FLAT_ITEMS], e[FLAT_ITEMS], row[FLAT_ITEMS];
    load_items_nt(pc, i0, n, vec_ok, -1, c);
    load_items_nt(ps, i0, n, vec_ok, 0, s);
    load_items_nt(pe, i0, n, vec_ok, 0, e);
    if (probe_ids) load_items(probe_ids, i0, n, vec_ok, 0, row);
    else {
#pragma unroll
        for (int k = 0; k < FLAT_ITEMS; ++k) row[k] = (int32_t)(i0 + k);
    }
    int lo[FLAT_ITEMS], cn[FLAT_ITEMS];
    long long tsum = 0;
#pragma unroll
    for (int k = 0; k < FLAT_ITEMS; ++k) {
        flat_range<STRICT>(ix, c[k], s[k], e[k], i0 + k < n, lo[k], cn[k]);
        // a candidate range this long is not a dense window, it is a window kept open by a few long rows (a contig-wide one):
        // testing every row of it is the cliff hier_walk exists for -- the probe is dropped, the flag makes the host redo the
        // call with the window kernels
        if (cn[k] > FLAT_MAX_CAND) { cn[k] = 0; atomicOr(&state[2], 1ull); }
        tsum += cn[k];
    }
    long long T;
    long long off[FLAT_ITEMS];
    off[0] = block_exclusive_scan(tsum, SumOp(), 0ll, lds_ll, &T);
    if (T == 0) return;                                    // uniform
#pragma unroll
    for (int k = 1; k < FLAT_ITEMS; ++k) off[k] = off[k - 1] + cn[k - 1];
#pragma unroll
    for (int k = 0; k < FLAT_ITEMS; ++k) {
        const int q = threadIdx.x * FLAT_ITEMS + k;
        l_lo[q] = lo[k]; l_off[q] = (uint32_t)off[k]; l_q[q] = make_int2(s[k], e[k]); l_row[q] = row[k];
    }
    // (the first barrier inside flat_chunk publishes these)
    const bool single = T <= (long long)FLAT_CH;
    long long wcnt = 0;                                    // matches of this wavefront over the whole tile
    if (single) {
        const int nC = (int)T;
        const int per = ((nC + FLAT_THREADS - 1) / FLAT_THREADS) * kWave;
        wcnt = flat_chunk<STRICT, true>(ix, 0ll, nC, per, off, cn, l_lo, l_off, l_q, marks, st_b, lds_i);
    } else {
        // A tile with more candidates than one chunk must know its number of matches before it can reserve its
        // output range.  With the end order at hand (rank_counts) that number comes from the two-rank formula of
        // count_overlaps -- #{start (<) q.end} - #{!(q.start (<) end)}: two table reads per probe instead of a sweep
        // over all candidates -- whenever the formula is exact (no inverted build row, no degenerate probe).
        bool formula = rank_counts != 0 && ix.flags[0] == 0;
        long long mine = 0;
        if (formula) {
            bool valid[FLAT_ITEMS], bad = false;
            int a[FLAT_ITEMS], b[FLAT_ITEMS], hi[FLAT_ITEMS], r[FLAT_ITEMS];
#pragma unroll
            for (int k = 0; k < FLAT_ITEMS; ++k) {
                valid[k] = i0 + k < n;
                bad |= valid[k] && (STRICT ? (s[k] >= e[k]) : (s[k] > e[k]));
            }
            bound_hi_tab4<STRICT>(ix, c, valid, e, a, b, hi);
            bound_r_tab4<STRICT>(ix, c, valid, s, r);
#pragma unroll
            for (int k = 0; k < FLAT_ITEMS; ++k) if (valid[k] && b[k] > a[k]) mine += (long long)hi[k] - (long long)r[k];
            formula = __syncthreads_or(bad ? 1 : 0) == 0;      // a degenerate probe anywhere in the tile: count by sweep
        }
        if (formula) {
            // reduce over the wavefront so that lane 0 holds the wavefront's total
#pragma unroll
            for (int d = kWave / 2; d > 0; d >>= 1) mine += __shfl_down(mine, d, kWave);
            wcnt = mine;                                        // meaningful in lane 0 only (that is what is stored)
        } else {
            for (long long c0 = 0; c0 < T; c0 += FLAT_CH) {
                const int nC = (int)((T - c0) < (long long)FLAT_CH ? (T - c0) : (long long)FLAT_CH);
                const int per = ((nC + FLAT_THREADS - 1) / FLAT_THREADS) * kWave;
                wcnt += flat_chunk<STRICT, false>(ix, c0, nC, per, off, cn, l_lo, l_off, l_q, marks, st_b, lds_i);
            }
        }
    }
    if (lane == 0) s_wtot[w] = wcnt;
    __syncthreads();
    long long pre = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < FLAT_THREADS / kWave; ++k) { const long long x = s_wtot[k]; if (k < w) pre += x; tot += x; }
    if (threadIdx.x == 0) {
        const long long base = tot ? (long long)atomicAdd(&state[0], (unsigned long long)tot) : 0ll;
        if (base + tot > capacity) { atomicExch(&state[1], 1ull); s_base = -1; }
        else s_base = base;
    }
    __syncthreads();
    const long long tbase = s_base;
    if (tbase < 0 || tot == 0) return;                     // uniform
    if (single) {
        const int nC = (int)T;
        const int per = ((nC + FLAT_THREADS - 1) / FLAT_THREADS) * kWave;
        const int wb = w * per;
        for (int j = lane; j < (int)wcnt; j += kWave) {
            __builtin_nontemporal_store(l_row[marks[wb + j]], out_probe + tbase + pre + j);
            __builtin_nontemporal_store(st_b[wb + j], out_build + tbase + pre + j);
        }
        return;
    }
    long long running = tbase;
    for (long long c0 = 0; c0 < T; c0 += FLAT_CH) {
        const int nC = (int)((T - c0) < (long long)FLAT_CH ? (T - c0) : (long long)FLAT_CH);
        const int per = ((nC + FLAT_THREADS - 1) / FLAT_THREADS) * kWave;
        const int cnt = flat_chunk<STRICT, true>(ix, c0, nC, per, off, cn, l_lo, l_off, l_q, marks, st_b, lds_i);
        if (lane == 0) s_wtot[w] = cnt;                    // (readers of the previous values have passed a barrier inside flat_chunk)
        __syncthreads();
        long long cpre = 0, ctot = 0;
#pragma unroll
        for (int k = 0; k < FLAT_THREADS / kWave; ++k) { const long long x = s_wtot[k]; if (k < w) cpre += x; ctot += x; }
        const int wb = w * per;
        for (int j = lane; j < cnt; j += kWave) {
            __builtin_nontemporal_store(l_row[marks[wb + j]], out_probe + running + cpre + j);
            __builtin_nontemporal_store(st_b[wb + j], out_build + running + cpre + j);
        }
        running += ctot;
    }
}

}  // namespace ivj
