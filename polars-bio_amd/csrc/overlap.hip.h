// overlap.hip.h -- pb.overlap kernels: count -> fill (deterministic), fused single pass, dense fill.
#pragma once
#include "index_view.hip.h"

namespace ivj {

// ------------------------------------------------------------------ overlap: count -> fill

// ---- shared bodies of the count / fill / fused kernels -------------------------------------------

constexpr int COOP_MIN = 16;            // matches a 64-row cooperative step must find for the window to stay with the wavefront

// For the PROBE_ITEMS probes of this thread: hi-bound through the table, then the window below hi
// as a 32-row match mask (x = mask) or -- window longer than 32 rows -- an exact count made by
// the whole wavefront (x = count, sign bit of hi set).  cnt = number of matches.
template <bool STRICT>
__device__ __forceinline__ void probe_windows(const IndexView& ix, const int32_t (&c)[PROBE_ITEMS],
                                              const int32_t (&s)[PROBE_ITEMS], const int32_t (&e)[PROBE_ITEMS],
                                              const bool (&valid)[PROBE_ITEMS], int (&hi)[PROBE_ITEMS],
                                              int (&x)[PROBE_ITEMS], int (&cnt)[PROBE_ITEMS]) {
    int a[PROBE_ITEMS], b[PROBE_ITEMS];
    bound_hi_tab4<STRICT>(ix, c, valid, e, a, b, hi);
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int k = 0; k < PROBE_ITEMS; ++k) {
        uint32_t mask; int cn;
        const bool small = window_mask<STRICT>(ix, a[k], hi[k], s[k], mask, cn);
        x[k] = (int)mask;
        // A window longer than the mask.  Dense windows (deeply nested / long build rows under many probes) are counted by the
        // whole wavefront, 64 rows per step with one coalesced read -- as long as a step still finds COOP_MIN matches; a window
        // that runs on with few matches (kept open by a handful of long rows, a contig-wide one) is handed to its lane, which
        // finds the remaining matches over the block maxima of the ends (hier_walk): a few reads per match however far down
        // they lie.  The emission (emit_tile_rows) replays exactly these decisions.
        int cont = -1;                                         // this lane's window: first row the cooperative steps did not cover
        unsigned long long todo = __ballot(!small);
        while (todo) {
            int src[4], ca[4], chi[4]; int32_t cqs[4]; int2 v0[4], v1[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                src[t] = todo ? __ffsll((long long)todo) - 1 : -1;
                if (todo) todo &= todo - 1;
                const int sl = src[t] < 0 ? 0 : src[t];
                ca[t] = __shfl(a[k], sl, kWave); chi[t] = __shfl(hi[k], sl, kWave); cqs[t] = __shfl(s[k], sl, kWave);
                if (src[t] < 0) { ca[t] = 0; chi[t] = 0; }
                const int p = chi[t] - 1 - lane;
                v0[t] = (p >= ca[t]) ? ix.ep[p] : make_int2(0, 0);
                v1[t] = (p - kWave >= ca[t]) ? ix.ep[p - kWave] : make_int2(0, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (src[t] < 0) continue;                      // uniform
                int cc = 0, rest = -1;
                int2 v = v0[t];
                int step = 0;
                for (int p0 = chi[t] - 1; p0 >= ca[t]; p0 -= kWave, ++step) {
                    const int p = p0 - lane;
                    if (step == 1) v = v1[t];
                    else if (step > 1) v = (p >= ca[t]) ? ix.ep[p] : make_int2(0, 0);
                    const bool pass = p >= ca[t] && lt_op<STRICT>(cqs[t], v.y);
                    const bool match = pass && lt_op<STRICT>(cqs[t], v.x);
                    const int nm = (int)__popcll(__ballot(match));
                    cc += nm;
                    if (__popcll(__ballot(pass)) < kWave) break;
                    if (nm < COOP_MIN) { rest = p0 - kWave; break; }
                }
                if (lane == src[t]) { cn = cc; cont = rest; }
            }
        }
        if (!small) {
            if (cont >= a[k]) hier_walk<STRICT>(ix.hier, [&](int p) { return ix.ep[p]; }, a[k], cont, s[k], [&](int) { ++cn; return true; });
            x[k] = cn;
        }
        if (!small) hi[k] |= (int)0x80000000;      // flag: x is a count, the emission rescans
        cnt[k] = cn;
    }
}

// Emission of one tile.  The pairs of a tile occupy ONE contiguous output range starting at
// `tbase`; they are compacted in LDS (windows of FILL_STAGE pairs, usually one) at their
// tile-local offset and copied out with fully coalesced stores.  Mask probes: bit j <=> row
// hi-1-j, ascending (start,row) order = descending j.  Long windows (flagged): the whole wavefront
// rescans 64 rows per step; the f-th match from the top of the window owns slot end-1-f, so a
// ballot + popcount of the lower lanes gives every matching lane its slot.
constexpr int FILL_STAGE = 3072;

struct GlobalRow {                      // staged build value = original build row of sorted position p
    const int32_t* b_row;
    __device__ __forceinline__ int32_t operator()(int p) const { return b_row[p]; }
};
struct PositionRow {                    // staged build value = the sorted position itself (resolved at copy-out)
    __device__ __forceinline__ int32_t operator()(int p) const { return p; }
};
struct PairPosOut {                     // staged build value = sorted position; the build row is read at copy-out
    int32_t* __restrict__ out_probe;
    int32_t* __restrict__ out_build;
    const int32_t* __restrict__ b_row;
    __device__ __forceinline__ void operator()(long long o, int32_t a, int32_t b) const {
        __builtin_nontemporal_store(a, out_probe + o);            // the result is a write-once stream
        __builtin_nontemporal_store(b_row[b], out_build + o);
    }
};
// copy-out of one staged pair (a = staged probe value, b = staged build value) to output slot o
struct PairOut {
    int32_t* __restrict__ out_probe;
    int32_t* __restrict__ out_build;
    __device__ __forceinline__ void operator()(long long o, int32_t a, int32_t b) const { out_probe[o] = a; out_build[o] = b; }
};

template <bool STRICT, int THREADS, int STAGE, class RowOf, class Out, int N>
__device__ __forceinline__ void emit_tile_rows(const IndexView& ix, const RowOf& rowof, const Out& out, const int32_t (&hi)[N],
                                               const int32_t (&x)[N], const int32_t (&cnt)[N],
                                               const int32_t (&row)[N], const int32_t (&qs)[N],
                                               long long loc0, long long tot, long long tbase, int32_t* st_p, int32_t* st_b) {
    const int lane = threadIdx.x & (kWave - 1);
    const unsigned long long lt_lanes = (1ull << lane) - 1ull;
    for (long long w0 = 0; w0 < tot; w0 += STAGE) {
        const long long w1 = w0 + STAGE;
        long long off = loc0;                                  // tile-local offset of the current probe
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const long long end = off + cnt[k];
            const bool in_win = cnt[k] != 0 && end > w0 && off < w1;
            if (in_win && hi[k] >= 0) {
                uint32_t m = (uint32_t)x[k];
                long long o = off;
                while (m) {
                    const int j = 31 - __clz(m);
                    m &= ~(1u << j);
                    if (o >= w0 && o < w1) {
                        st_p[o - w0] = row[k];
                        st_b[o - w0] = rowof(hi[k] - 1 - j);
                    }
                    ++o;
                }
            }
            // long windows: the decisions of the count replayed (probe_windows): cooperative steps of 64 rows while a step finds
            // COOP_MIN matches, the f-th match from the top of the window owns slot end - 1 - f (ballot + popcount of the lower
            // lanes); what is left belongs to the lane's own walk over the block maxima
            int cont = -1, done = 0;
            unsigned long long todo = __ballot(in_win && hi[k] < 0);
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int h = __shfl(hi[k], src, kWave) & 0x7fffffff;
                const int c = __shfl(cnt[k], src, kWave);
                const int32_t cqs = __shfl(qs[k], src, kWave);
                const int32_t crow = __shfl(row[k], src, kWave);
                const long long cend = ((long long)__shfl((int)(end >> 32), src, kWave) << 32) |
                                       (unsigned long long)(unsigned int)__shfl((int)(end & 0xffffffffll), src, kWave);
                int found = 0, rest = -1;
                for (int p0 = h - 1; found < c && p0 >= 0 && cend - found > w0; p0 -= kWave) {
                    const int p = p0 - lane;
                    int2 v = make_int2(0, 0);
                    int32_t br = 0;
                    if (p >= 0) { v = ix.ep[p]; br = rowof(p); }
                    const bool m = p >= 0 && lt_op<STRICT>(cqs, v.x);
                    const unsigned long long mm = __ballot(m);
                    if (m) {
                        // rows below the window (or of the previous contig) rank past the c-th match
                        const long long o = cend - 1 - found - (long long)__popcll(mm & lt_lanes);
                        if (o >= w0 && o < w1 && o >= cend - c) { st_p[o - w0] = crow; st_b[o - w0] = br; }
                    }
                    const int nm = (int)__popcll(mm);
                    found += nm;
                    if (found < c && nm < COOP_MIN) { rest = p0 - kWave; break; }
                }
                if (lane == src) { cont = rest; done = found; }
            }
            if (cont >= 0) {
                // (the first cnt matches going down are the window's: the walk needs no lower bound, it stops there or below
                // the staging window)
                long long o = end - 1 - done;
                if (o >= off && o >= w0)
                    hier_walk<STRICT>(ix.hier, [&](int p) { return ix.ep[p]; }, 0, cont, qs[k], [&](int p) {
                        if (o < w1) { st_p[o - w0] = row[k]; st_b[o - w0] = rowof(p); }
                        --o;
                        return o >= off && o >= w0;
                    });
            }
            off = end;
        }
        __syncthreads();
        const int t = (int)((tot - w0) < (long long)STAGE ? (tot - w0) : (long long)STAGE);
        for (int i = threadIdx.x; i < t; i += THREADS) out(tbase + w0 + i, st_p[i], st_b[i]);
        __syncthreads();
    }
}

template <bool STRICT>
__device__ __forceinline__ void emit_tile(const IndexView& ix, const int32_t (&hi)[PROBE_ITEMS],
                                          const int32_t (&x)[PROBE_ITEMS], const int32_t (&cnt)[PROBE_ITEMS],
                                          const int32_t (&row)[PROBE_ITEMS], const int32_t (&qs)[PROBE_ITEMS],
                                          long long loc0, long long tot, long long tbase, int32_t* st_p, int32_t* st_b,
                                          int32_t* __restrict__ out_probe, int32_t* __restrict__ out_build) {
    emit_tile_rows<STRICT, PROBE_THREADS, FILL_STAGE>(ix, PositionRow{}, PairPosOut{out_probe, out_build, ix.b_row}, hi, x, cnt, row, qs, loc0, tot,
                                                       tbase, st_p, st_b);
}

// Pass 1.  One workgroup = PROBE_TILE probes, PROBE_ITEMS consecutive probes per thread.
// Writes hi[i] and the 32-row match mask of the window below hi (or, flagged in the sign bit of
// hi, the exact count of a longer window) so the fill pass neither searches nor rescans.
template <bool STRICT>
__global__ __launch_bounds__(PROBE_THREADS) void k_overlap_count(IndexView ix, const int32_t* __restrict__ pc,
                                                                 const int32_t* __restrict__ ps,
                                                                 const int32_t* __restrict__ pe, int64_t n, bool vec_ok,
                                                                 int32_t* __restrict__ hi_out, int32_t* __restrict__ cnt_out,
                                                                 long long* __restrict__ tile_tot) {
    __shared__ long long lds[PROBE_THREADS / kWave];
    const long long ntiles = (n + PROBE_TILE - 1) / PROBE_TILE;
    const long long tile = xcd_tile64(blockIdx.x, ntiles);
    if (tile >= ntiles) return;                            // uniform
    const int64_t i0 = (int64_t)tile * PROBE_TILE + (int64_t)threadIdx.x * PROBE_ITEMS;
    int32_t c[PROBE_ITEMS], s[PROBE_ITEMS], e[PROBE_ITEMS];
    load_items_nt(pc, i0, n, vec_ok, -1, c);
    load_items_nt(ps, i0, n, vec_ok, 0, s);
    load_items_nt(pe, i0, n, vec_ok, 0, e);
    int hi[PROBE_ITEMS], x[PROBE_ITEMS], cnt[PROBE_ITEMS];
    bool valid[PROBE_ITEMS];
#pragma unroll
    for (int k = 0; k < PROBE_ITEMS; ++k) valid[k] = i0 + k < n;
    probe_windows<STRICT>(ix, c, s, e, valid, hi, x, cnt);
    long long tsum = 0;
#pragma unroll
    for (int k = 0; k < PROBE_ITEMS; ++k) tsum += cnt[k];
    store_items(hi_out, i0, n, vec_ok, hi);
    store_items(cnt_out, i0, n, vec_ok, x);
    long long tot;
    block_exclusive_scan(tsum, SumOp(), 0ll, lds, &tot);
    if (threadIdx.x == 0) tile_tot[tile] = tot;
}

// Pass 2.  tile_base = exclusive scan of tile_tot.
template <bool STRICT>
__global__ __launch_bounds__(PROBE_THREADS, 5) void k_overlap_fill(IndexView ix, const int32_t* __restrict__ ps, int64_t n,
                                                                bool vec_ok, const int32_t* __restrict__ hi_in,
                                                                const int32_t* __restrict__ cnt_in,
                                                                const long long* __restrict__ tile_base,
                                                                const int32_t* __restrict__ probe_ids,
                                                                int32_t* __restrict__ out_probe,
                                                                int32_t* __restrict__ out_build) {
    __shared__ long long lds[PROBE_THREADS / kWave];
    __shared__ int32_t st_p[FILL_STAGE];
    __shared__ int32_t st_b[FILL_STAGE];
    const long long ntiles = (n + PROBE_TILE - 1) / PROBE_TILE;
    const long long tile = xcd_tile64(blockIdx.x, ntiles);
    if (tile >= ntiles) return;                            // uniform
    const int64_t i0 = (int64_t)tile * PROBE_TILE + (int64_t)threadIdx.x * PROBE_ITEMS;
    int32_t hi[PROBE_ITEMS], x[PROBE_ITEMS], cnt[PROBE_ITEMS], row[PROBE_ITEMS], qs[PROBE_ITEMS];
    load_items(hi_in, i0, n, vec_ok, 0, hi);
    load_items(cnt_in, i0, n, vec_ok, 0, x);
    long long tsum = 0;
#pragma unroll
    for (int k = 0; k < PROBE_ITEMS; ++k) {
        cnt[k] = hi[k] < 0 ? x[k] : __popc((uint32_t)x[k]);
        tsum += cnt[k];
        row[k] = (int32_t)(i0 + k);
    }
    if (probe_ids && tsum) {
#pragma unroll
        for (int k = 0; k < PROBE_ITEMS; ++k) if (cnt[k]) row[k] = probe_ids[i0 + k];
    }
#pragma unroll
    for (int k = 0; k < PROBE_ITEMS; ++k) qs[k] = (hi[k] < 0 && cnt[k] != 0) ? ps[i0 + k] : 0;
    long long tot;
    const long long loc0 = block_exclusive_scan(tsum, SumOp(), 0ll, lds, &tot);
    emit_tile<STRICT>(ix, hi, x, cnt, row, qs, loc0, tot, tile_base[tile], st_p, st_b, out_probe, out_build);
}

// Fused single pass (count + fill) for callers that bring an output buffer of known capacity
// (steady-state / streaming use: the previous batch sized it).  Each tile reserves its output range
// with ONE 64-bit atomicAdd on a cursor, so no tile waits for another and nothing is written to or
// re-read from HBM between counting and emitting.  Tile ranges land in reservation order: the
// pairs of one probe row stay contiguous and ordered, the order of tiles is not reproducible from
// run to run (the two-pass path is the deterministic one).  state[0] = cursor (= total on exit),
// state[1] = 1 when the capacity was exceeded (nothing is written past it).
template <bool STRICT>
__global__ __launch_bounds__(PROBE_THREADS, 5) void k_overlap_fused(IndexView ix, const int32_t* __restrict__ pc,
                                                                 const int32_t* __restrict__ ps,
                                                                 const int32_t* __restrict__ pe,
                                                                 const int32_t* __restrict__ probe_ids, int64_t n,
                                                                 bool vec_ok, long long capacity,
                                                                 unsigned long long* __restrict__ state,
                                                                 int32_t* __restrict__ out_probe,
                                                                 int32_t* __restrict__ out_build) {
    __shared__ long long lds[PROBE_THREADS / kWave];
    __shared__ long long s_base;
    __shared__ int32_t st_p[FILL_STAGE];
    __shared__ int32_t st_b[FILL_STAGE];
    const long long ntiles = (n + PROBE_TILE - 1) / PROBE_TILE;
    const long long tile = xcd_tile64(blockIdx.x, ntiles);
    if (tile >= ntiles) return;                            // uniform
    const int64_t i0 = (int64_t)tile * PROBE_TILE + (int64_t)threadIdx.x * PROBE_ITEMS;
    int32_t c[PROBE_ITEMS], s[PROBE_ITEMS], e[PROBE_ITEMS];
    load_items_nt(pc, i0, n, vec_ok, -1, c);
    load_items_nt(ps, i0, n, vec_ok, 0, s);
    load_items_nt(pe, i0, n, vec_ok, 0, e);
    int hi[PROBE_ITEMS], x[PROBE_ITEMS], cnt[PROBE_ITEMS], row[PROBE_ITEMS];
    bool valid[PROBE_ITEMS];
#pragma unroll
    for (int k = 0; k < PROBE_ITEMS; ++k) valid[k] = i0 + k < n;
    probe_windows<STRICT>(ix, c, s, e, valid, hi, x, cnt);
    long long tsum = 0;
#pragma unroll
    for (int k = 0; k < PROBE_ITEMS; ++k) { tsum += cnt[k]; row[k] = (int32_t)(i0 + k); }
    if (probe_ids && tsum) {
#pragma unroll
        for (int k = 0; k < PROBE_ITEMS; ++k) if (cnt[k]) row[k] = probe_ids[i0 + k];
    }
    long long tot;
    const long long loc0 = block_exclusive_scan(tsum, SumOp(), 0ll, lds, &tot);
    if (threadIdx.x == 0) {
        const long long base = tot ? (long long)atomicAdd(&state[0], (unsigned long long)tot) : 0ll;
        if (base + tot > capacity) { atomicExch(&state[1], 1ull); s_base = -1; }
        else s_base = base;
    }
    __syncthreads();
    const long long tbase = s_base;
    if (tbase < 0 || tot == 0) return;                     // uniform
    emit_tile<STRICT>(ix, hi, x, cnt, row, s, loc0, tot, tbase, st_p, st_b, out_probe, out_build);
}

// ---- fused join + row materialisation ---------------------------------
// The joined ROWS leave the kernel, not just the index pairs: per pair the probe row, the build row,
// the contig id and start/end of both sides (reference: the SELECT over the joined batches,
// src/operation.rs:272-301, for the key columns).  A separate gather pass over the finished pair list
// has to fetch three probe columns at random over the whole probe side (10 ms for config 3); here
// the probe values never leave the workgroup (staged pair = tile-local probe slot -> LDS tables of the
// tile) and the build values come with ONE 16-byte rec4 read per pair at copy-out, where consecutive
// lanes hold consecutive rows of the same probe.
constexpr int ROWS_STAGE = 2048;

struct RowColumns {                     // device pointers, any may be null (column skipped)
    int32_t *probe_idx, *build_idx, *contig, *start_1, *end_1, *start_2, *end_2;
};
struct RowsOut {
    RowColumns c;
    const int4* __restrict__ rec4;      // {start, end, build row, pmax} per sorted position
    const int32_t *l_row, *l_c, *l_s, *l_e;   // LDS tables of the tile, by tile-local probe slot
    __device__ __forceinline__ void operator()(long long o, int32_t q, int32_t pos) const {
        const int4 r = rec4[pos];
        // write-once result streams: non-temporal stores keep them from displacing the index slices in the L2
        if (c.probe_idx) __builtin_nontemporal_store(l_row[q], c.probe_idx + o);
        if (c.build_idx) __builtin_nontemporal_store(r.z, c.build_idx + o);
        if (c.contig) __builtin_nontemporal_store(l_c[q], c.contig + o);
        if (c.start_1) __builtin_nontemporal_store(l_s[q], c.start_1 + o);
        if (c.end_1) __builtin_nontemporal_store(l_e[q], c.end_1 + o);
        if (c.start_2) __builtin_nontemporal_store(r.x, c.start_2 + o);
        if (c.end_2) __builtin_nontemporal_store(r.y, c.end_2 + o);
    }
};

template <bool STRICT>
__global__ __launch_bounds__(PROBE_THREADS, 4) void k_overlap_fused_rows(IndexView ix, const int32_t* __restrict__ pc,
                                                                       const int32_t* __restrict__ ps,
                                                                       const int32_t* __restrict__ pe,
                                                                       const int32_t* __restrict__ probe_ids, int64_t n,
                                                                       bool vec_ok, long long capacity,
                                                                       unsigned long long* __restrict__ state, RowColumns cols) {
    __shared__ long long lds[PROBE_THREADS / kWave];
    __shared__ long long s_base;
    __shared__ int32_t st_p[ROWS_STAGE];
    __shared__ int32_t st_b[ROWS_STAGE];
    __shared__ int32_t l_row[PROBE_TILE], l_c[PROBE_TILE], l_s[PROBE_TILE], l_e[PROBE_TILE];
    const long long ntiles = (n + PROBE_TILE - 1) / PROBE_TILE;
    const long long tile = xcd_tile64(blockIdx.x, ntiles);
    if (tile >= ntiles) return;                            // uniform
    const int64_t i0 = (int64_t)tile * PROBE_TILE + (int64_t)threadIdx.x * PROBE_ITEMS;
    int32_t c[PROBE_ITEMS], s[PROBE_ITEMS], e[PROBE_ITEMS], row[PROBE_ITEMS];
    load_items_nt(pc, i0, n, vec_ok, -1, c);
    load_items_nt(ps, i0, n, vec_ok, 0, s);
    load_items_nt(pe, i0, n, vec_ok, 0, e);
    if (probe_ids) load_items(probe_ids, i0, n, vec_ok, 0, row);
    else {
#pragma unroll
        for (int k = 0; k < PROBE_ITEMS; ++k) row[k] = (int32_t)(i0 + k);
    }
    int32_t slot[PROBE_ITEMS];
#pragma unroll
    for (int k = 0; k < PROBE_ITEMS; ++k) {
        slot[k] = threadIdx.x * PROBE_ITEMS + k;
        l_row[slot[k]] = row[k]; l_c[slot[k]] = c[k]; l_s[slot[k]] = s[k]; l_e[slot[k]] = e[k];
    }
    int hi[PROBE_ITEMS], x[PROBE_ITEMS], cnt[PROBE_ITEMS];
    bool valid[PROBE_ITEMS];
#pragma unroll
    for (int k = 0; k < PROBE_ITEMS; ++k) valid[k] = i0 + k < n;
    probe_windows<STRICT>(ix, c, s, e, valid, hi, x, cnt);
    long long tsum = 0;
#pragma unroll
    for (int k = 0; k < PROBE_ITEMS; ++k) tsum += cnt[k];
    long long tot;
    const long long loc0 = block_exclusive_scan(tsum, SumOp(), 0ll, lds, &tot);     // (its barriers publish the LDS tables)
    if (threadIdx.x == 0) {
        const long long base = tot ? (long long)atomicAdd(&state[0], (unsigned long long)tot) : 0ll;
        if (base + tot > capacity) { atomicExch(&state[1], 1ull); s_base = -1; }
        else s_base = base;
    }
    __syncthreads();
    const long long tbase = s_base;
    if (tbase < 0 || tot == 0) return;                     // uniform
    emit_tile_rows<STRICT, PROBE_THREADS, ROWS_STAGE>(ix, PositionRow{}, RowsOut{cols, ix.rec4, l_row, l_c, l_s, l_e}, hi, x, cnt, slot, s, loc0,
                                                       tot, tbase, st_p, st_b);
}

// Pass 2 for dense results (many pairs per probe).  Same tiles, same output layout as
// k_overlap_fill, but the probes of a tile are first parked in LDS and every output window is
// shared out over ALL wavefronts of the workgroup (probe q of the window goes to wavefront
// q mod 4), because in a dense tile one window covers only a few dozen consecutive probes -- all
// owned by one wavefront in the per-lane scheme.  A wavefront emits one probe at a time: a mask
// probe with one lane per mask bit, a long window with 64 rows per step (ballot + popcount of the
// lower lanes = slot), (end,pmax) and build row of 128 rows requested up front.
constexpr int DENSE_STAGE = 2048;

template <bool STRICT, int N>
__global__ __launch_bounds__(PROBE_THREADS) void k_overlap_fill_dense(IndexView ix, const int32_t* __restrict__ ps, int64_t n,
                                                                      bool vec_ok, const int32_t* __restrict__ hi_in,
                                                                      const int32_t* __restrict__ cnt_in,
                                                                      const long long* __restrict__ tile_base,
                                                                      const int32_t* __restrict__ probe_ids,
                                                                      int32_t* __restrict__ out_probe,
                                                                      int32_t* __restrict__ out_build) {
    __shared__ long long lds[PROBE_THREADS / kWave];
    __shared__ int32_t st_p[DENSE_STAGE];
    __shared__ int32_t st_b[DENSE_STAGE];
    __shared__ int32_t l_hi[(PROBE_THREADS * N)], l_x[(PROBE_THREADS * N)], l_qs[(PROBE_THREADS * N)], l_row[(PROBE_THREADS * N)];
    __shared__ long long l_off[(PROBE_THREADS * N) + 1];
    const int64_t i0 = (int64_t)blockIdx.x * (PROBE_THREADS * N) + (int64_t)threadIdx.x * N;
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    const unsigned long long lt_lanes = (1ull << lane) - 1ull;
    {
        int32_t hi[N], x[N];
        load_items(hi_in, i0, n, vec_ok, 0, hi);
        load_items(cnt_in, i0, n, vec_ok, 0, x);
        long long tsum = 0;
        int cnt[N];
#pragma unroll
        for (int k = 0; k < N; ++k) { cnt[k] = hi[k] < 0 ? x[k] : __popc((uint32_t)x[k]); tsum += cnt[k]; }
        long long tot0;
        long long off = block_exclusive_scan(tsum, SumOp(), 0ll, lds, &tot0);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int q = threadIdx.x * N + k;
            l_hi[q] = hi[k]; l_x[q] = x[k]; l_off[q] = off;
            l_qs[q] = (hi[k] < 0 && cnt[k] != 0) ? ps[i0 + k] : 0;
            l_row[q] = (cnt[k] != 0 && probe_ids) ? probe_ids[i0 + k] : (int32_t)(i0 + k);
            off += cnt[k];
        }
        if (threadIdx.x == PROBE_THREADS - 1) l_off[(PROBE_THREADS * N)] = off;
    }
    __syncthreads();
    const long long tot = l_off[(PROBE_THREADS * N)];
    const long long tbase = tile_base[blockIdx.x];
    for (long long w0 = 0; w0 < tot; w0 += DENSE_STAGE) {
        const long long w1 = w0 + DENSE_STAGE;
        // probes intersecting [w0,w1): f = last probe with off <= w0, l = first probe with off >= w1
        int f, l;
        { int lo = 0, hi = (PROBE_THREADS * N); while (lo < hi) { const int m = (lo + hi) >> 1; if (l_off[m] <= w0) lo = m + 1; else hi = m; } f = lo - 1; }
        { int lo = 0, hi = (PROBE_THREADS * N); while (lo < hi) { const int m = (lo + hi) >> 1; if (l_off[m] < w1) lo = m + 1; else hi = m; } l = lo; }
        for (int q = f + w; q < l; q += PROBE_THREADS / kWave) {       // wavefront-uniform
            const long long off = l_off[q], end = l_off[q + 1];
            const int c = (int)(end - off);
            if (c == 0) continue;
            const int32_t hi = l_hi[q], crow = l_row[q];
            if (hi >= 0) {
                // mask probe: lane j owns bit j; its slot = off + number of set bits above j
                const uint32_t x = (uint32_t)l_x[q];
                if (lane < 32 && ((x >> lane) & 1u)) {
                    const uint32_t above = lane == 31 ? 0u : (x & ~((2u << lane) - 1u));
                    const long long o = off + __popc(above);
                    if (o >= w0 && o < w1) { st_p[o - w0] = crow; st_b[o - w0] = ix.b_row[hi - 1 - lane]; }
                }
            } else {
                const int h = hi & 0x7fffffff;
                const int32_t cqs = l_qs[q];
                const int pa = h - 1 - lane, pb = pa - kWave;
                int2 va = make_int2(0, 0), vb = make_int2(0, 0);
                int32_t ra = 0, rb = 0;
                if (pa >= 0) { va = ix.ep[pa]; ra = ix.b_row[pa]; }
                if (pb >= 0) { vb = ix.ep[pb]; rb = ix.b_row[pb]; }
                int found = 0, step = 0;
                for (int p0 = h - 1; found < c && p0 >= 0 && end - found > w0; p0 -= kWave, ++step) {
                    const int p = p0 - lane;
                    int2 v; int32_t br;
                    if (step == 0) { v = va; br = ra; }
                    else if (step == 1) { v = vb; br = rb; }
                    else { v = make_int2(0, 0); br = 0; if (p >= 0) { v = ix.ep[p]; br = ix.b_row[p]; } }
                    const bool m = p >= 0 && lt_op<STRICT>(cqs, v.x);
                    const unsigned long long mm = __ballot(m);
                    if (m) {
                        const long long o = end - 1 - found - (long long)__popcll(mm & lt_lanes);
                        if (o >= w0 && o < w1 && o >= off) { st_p[o - w0] = crow; st_b[o - w0] = br; }
                    }
                    found += (int)__popcll(mm);
                }
            }
        }
        __syncthreads();
        const int t = (int)((tot - w0) < (long long)DENSE_STAGE ? (tot - w0) : (long long)DENSE_STAGE);
        for (int i = threadIdx.x; i < t; i += PROBE_THREADS) {
            __builtin_nontemporal_store(st_p[i], out_probe + tbase + w0 + i);
            __builtin_nontemporal_store(st_b[i], out_build + tbase + w0 + i);
        }
        __syncthreads();
    }
}

}  // namespace ivj
