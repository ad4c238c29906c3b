// probe.hip.h -- index finalisation and probe kernels of the interval join.
//
// HBM layout of the build index (sorted by (contig, start, row)):
//   b_start[Nb]  int32   start, the array the hi-bound search runs on
//   ep[Nb]       int2    (end, prefix-max of end inside the contig segment)
//   b_row[Nb]    int32   original build row
//   b_contig[Nb] int32   contig id in that order (rows outside the dictionary get n_contigs)
//   seg[n_contigs + 2]   segment offsets; seg[n_contigs] = number of valid rows
//   e_end / e_pos [Nb]   optional: ends sorted by (contig, end, position), and that position
//   cmeta[n_contigs]     per-contig {segment, min/max start, bin shift, table offset}
//   brec[2 Nb + 2 n_contigs] direct-address table over start, 16 B per bin: first position of the
//                        bin and the keys of the next three rows; about one build row per bin, so
//                        the hi-bound of a probe is ONE 16-byte gather (3 compares) instead of a
//                        log2(Nb)-step binary search of dependent gathers
//
// Predicate (polars_bio/range_op.py:75-84; src/option.rs:95-100):
//   STRICT: q.start <  b.end && b.start <  q.end      WEAK: <=
// For a probe q on contig c with segment [a,b):
//   hi = first p in [a,b) with !(b_start[p] (<) q.end)     -> every match has p < hi
//   matches = { p in [a,hi) : q.start (<) end[p] };  the prefix max bounds the backward scan:
//   stop at the first p (going down) with !(q.start (<) pmax[p]).
#pragma once
#include "radix_sort.hip.h"
#include "scan.hip.h"

namespace ivj {

constexpr int PROBE_THREADS = 256;
constexpr int PROBE_ITEMS = 4;   // probes per thread of the overlap count / fill / fused kernels
constexpr int PROBE_ITEMS_LAT = 2;   // nearest and the dense fill: shorter per-thread chains, full occupancy
constexpr int PROBE_TILE = PROBE_THREADS * PROBE_ITEMS;

struct IndexView {
    const int32_t* b_start;
    const int2* ep;
    const int32_t* b_row;
    const int32_t* seg;
    const int32_t* e_end;
    const int32_t* e_pos;
    const int32_t* flags;  // flags[0] != 0: some build row has start > end
    const int4* cmeta;     // per contig: {a, b, ulo, uhi} {shift, tb, 0, 0}  (two int4)
    const int4* brec;      // direct-address table: brec[tb + j] = {p0, key[p0], key[p0+1], key[p0+2]} with
                           // p0 = first position whose ustart >= ulo + (j << shift)
    const uint32_t* bins;  // the same table as plain first positions (4 B per bin): used instead of brec for
                           // small build sides, whose 4-byte tables + key arrays stay L2-resident
    const int4* cmeta_e;   // the same pair of structures over the end-sorted order (e_end)
    const int4* brec_e;
    const uint32_t* bins_e;
    int32_t use_rec;       // 1: gather 16-byte records, 0: 4-byte bins + bound search on the key array
    const int32_t* pargmax; // position of the first row that attains ep[p].y (prefix max) -- nearest only
    const int4* cmeta_j;    // count_overlaps: ONE bin grid per contig shared by the start- and the end-sorted order
    const int4* crec;       //   crec[slot] = {first start position, its start, first end position, its end} of the bin
    const int4* nrec;       // nearest: nrec[p] = {pmax[p-1], row of its argmax, start[p], end[p]} (left / right candidate of hi = p)
    int32_t n_contigs;
};

__device__ __forceinline__ uint32_t flip(int32_t v) { return (uint32_t)v ^ 0x80000000u; }
__device__ __forceinline__ int32_t unflip(uint32_t v) { return (int32_t)(v ^ 0x80000000u); }

template <bool STRICT>
__device__ __forceinline__ bool lt_op(int32_t x, int32_t y) { return STRICT ? (x < y) : (x <= y); }

__device__ __forceinline__ long long gap_dist(int32_t qs, int32_t qe, int32_t bs, int32_t be) {
    const long long d1 = (long long)bs - (long long)qe;
    const long long d2 = (long long)qs - (long long)be;
    return d1 > d2 ? d1 : d2;
}

__device__ __forceinline__ void seg_bounds(const IndexView& ix, int32_t c, bool valid, int& a, int& b) {
    if (valid && (uint32_t)c < (uint32_t)ix.n_contigs) { a = ix.seg[c]; b = ix.seg[c + 1]; }
    else { a = 0; b = 0; }
}

// first p in [lo,hi) with arr[p] >= x (OR_EQUAL=false: lower bound) / arr[p] > x (true: upper bound)
template <bool UPPER>
__device__ __forceinline__ int bsearch32(const int32_t* __restrict__ arr, int lo, int hi, int32_t x) {
    while (lo < hi) {
        const int m = lo + ((hi - lo) >> 1);
        const int32_t v = arr[m];
        const bool right = UPPER ? (v <= x) : (v < x);
        if (right) lo = m + 1; else hi = m;
    }
    return lo;
}
// same on the .y (prefix max) lane of ep
template <bool UPPER>
__device__ __forceinline__ int bsearch_pmax(const int2* __restrict__ ep, int lo, int hi, int32_t x) {
    while (lo < hi) {
        const int m = lo + ((hi - lo) >> 1);
        const int32_t v = ep[m].y;
        const bool right = UPPER ? (v <= x) : (v < x);
        if (right) lo = m + 1; else hi = m;
    }
    return lo;
}

// hi: first position whose start fails "start (<) q.end"
template <bool STRICT>
__device__ __forceinline__ int bound_hi(const IndexView& ix, int a, int b, int32_t qe) {
    return STRICT ? bsearch32<false>(ix.b_start, a, b, qe) : bsearch32<true>(ix.b_start, a, b, qe);
}
// lo: first position in [a,hi) whose prefix max satisfies "q.start (<) pmax"
template <bool STRICT>
__device__ __forceinline__ int bound_lo(const IndexView& ix, int a, int hi, int32_t qs) {
    return STRICT ? bsearch_pmax<true>(ix.ep, a, hi, qs) : bsearch_pmax<false>(ix.ep, a, hi, qs);
}
// r: first position of the end-sorted segment whose end satisfies "q.start (<) end"
template <bool STRICT>
__device__ __forceinline__ int bound_r(const IndexView& ix, int a, int b, int32_t qs) {
    return STRICT ? bsearch32<true>(ix.e_end, a, b, qs) : bsearch32<false>(ix.e_end, a, b, qs);
}

// Four interleaved hi-bound searches: the four gathers of a step are issued
// back to back, so a thread keeps four HBM/L2 requests in flight.
template <bool STRICT>
__device__ __forceinline__ void bound_hi4(const IndexView& ix, const int (&a)[PROBE_ITEMS], const int (&b)[PROBE_ITEMS],
                                          const int32_t (&qe)[PROBE_ITEMS], int (&out)[PROBE_ITEMS]) {
    int lo[PROBE_ITEMS], hi[PROBE_ITEMS];
#pragma unroll
    for (int k = 0; k < PROBE_ITEMS; ++k) { lo[k] = a[k]; hi[k] = b[k]; }
    for (;;) {
        bool any = false;
        int32_t v[PROBE_ITEMS];
        int m[PROBE_ITEMS];
#pragma unroll
        for (int k = 0; k < PROBE_ITEMS; ++k) {
            m[k] = lo[k] + ((hi[k] - lo[k]) >> 1);
            const bool act = lo[k] < hi[k];
            any |= act;
            v[k] = act ? ix.b_start[m[k]] : 0;
        }
        if (!any) break;
#pragma unroll
        for (int k = 0; k < PROBE_ITEMS; ++k) {
            if (lo[k] < hi[k]) {
                const bool right = STRICT ? (v[k] < qe[k]) : (v[k] <= qe[k]);
                if (right) lo[k] = m[k] + 1; else hi[k] = m[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PROBE_ITEMS; ++k) out[k] = lo[k];
}

// Lower bound through a direct-address table, four probes interleaved: out[k] = first position p
// of contig c[k]'s segment with flip(keys[p]) >= tu[k].  Targets are compared on the flipped
// (unsigned-ordered) coordinates in 64 bits, so negative coordinates and INT32_MAX + 1 need no
// special case.  One table read + a search over the rows of one bin.
template <int N>
__device__ __forceinline__ void lb_tab4(const int4* __restrict__ cmeta, const int4* __restrict__ brec,
                                        const uint32_t* __restrict__ bins, bool use_rec,
                                        const int32_t* __restrict__ keys, int32_t n_contigs,
                                        const int32_t (&c)[N], const bool (&valid)[N],
                                        const unsigned long long (&tu)[N],
                                        int (&a)[N], int (&b)[N], int (&out)[N]) {
    int4 m0[N], m1[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const bool ok = valid[k] && (uint32_t)c[k] < (uint32_t)n_contigs;
        if (ok) { m0[k] = cmeta[2 * c[k]]; m1[k] = cmeta[2 * c[k] + 1]; }
        else { m0[k] = make_int4(0, 0, 0, 0); m1[k] = make_int4(0, 0, 0, 0); }
    }
    int4 rec[N];
    uint32_t slot[N];
    bool inb[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        a[k] = m0[k].x; b[k] = m0[k].y;
        const uint32_t ulo = (uint32_t)m0[k].z, uhi = (uint32_t)m0[k].w;
        inb[k] = false; slot[k] = 0; rec[k] = make_int4(0, 0, 0, 0);
        if (b[k] <= a[k] || tu[k] <= ulo) out[k] = a[k];
        else if (tu[k] > uhi) out[k] = b[k];
        else {
            inb[k] = true;
            slot[k] = (uint32_t)m1[k].y + (((uint32_t)tu[k] - ulo) >> m1[k].x);
            if (use_rec) rec[k] = brec[slot[k]];
            else { rec[k].x = (int)bins[slot[k]]; rec[k].y = (int)bins[slot[k] + 1]; }
        }
    }
    if (use_rec) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (!inb[k]) continue;
            // rows p0, p0+1, p0+2 of the bin (or later bins / a sentinel past the segment): keys
            // ascend, so the number of leading keys below the target is the offset of the bound
            const bool n0 = (unsigned long long)flip(rec[k].y) < tu[k];
            const bool n1 = n0 && (unsigned long long)flip(rec[k].z) < tu[k];
            const bool n2 = n1 && (unsigned long long)flip(rec[k].w) < tu[k];
            int lo = rec[k].x + (n0 ? 1 : 0) + (n1 ? 1 : 0) + (n2 ? 1 : 0);
            if (n2) {
                // crowded bin: finish with a bound search up to the first row of the next bin
                int hi = brec[slot[k] + 1].x;
                while (lo < hi) {
                    const int m = lo + ((hi - lo) >> 1);
                    if ((unsigned long long)flip(keys[m]) < tu[k]) lo = m + 1; else hi = m;
                }
            }
            out[k] = lo;
        }
    } else {
        // four interleaved bound searches over the rows of one bin each
        int lo[N], hi[N];
#pragma unroll
        for (int k = 0; k < N; ++k) { lo[k] = inb[k] ? rec[k].x : 0; hi[k] = inb[k] ? rec[k].y : 0; }
        for (;;) {
            bool any = false;
            int32_t v[N];
            int m[N];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                m[k] = lo[k] + ((hi[k] - lo[k]) >> 1);
                const bool act = lo[k] < hi[k];
                any |= act;
                v[k] = act ? keys[m[k]] : 0;
            }
            if (!any) break;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                if (lo[k] < hi[k]) {
                    if ((unsigned long long)flip(v[k]) < tu[k]) lo[k] = m[k] + 1; else hi[k] = m[k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < N; ++k) if (inb[k]) out[k] = lo[k];
    }
}

// hi = first position whose start fails "start (<) q.end": first start >= q.end (STRICT) / > q.end (WEAK)
template <bool STRICT, int N>
__device__ __forceinline__ void bound_hi_tab4(const IndexView& ix, const int32_t (&c)[N],
                                              const bool (&valid)[N], const int32_t (&qe)[N],
                                              int (&a)[N], int (&b)[N], int (&out)[N]) {
    unsigned long long tu[N];
#pragma unroll
    for (int k = 0; k < N; ++k) tu[k] = (unsigned long long)flip(qe[k]) + (STRICT ? 0ull : 1ull);
    lb_tab4(ix.cmeta, ix.brec, ix.bins, ix.use_rec != 0, ix.b_start, ix.n_contigs, c, valid, tu, a, b, out);
}
// r = first position of the end-sorted segment whose end satisfies "q.start (<) end":
// first end > q.start (STRICT) / >= q.start (WEAK)
template <bool STRICT, int N>
__device__ __forceinline__ void bound_r_tab4(const IndexView& ix, const int32_t (&c)[N],
                                             const bool (&valid)[N], const int32_t (&qs)[N],
                                             int (&out)[N]) {
    unsigned long long tu[N];
    int a[N], b[N];
#pragma unroll
    for (int k = 0; k < N; ++k) tu[k] = (unsigned long long)flip(qs[k]) + (STRICT ? 1ull : 0ull);
    lb_tab4(ix.cmeta_e, ix.brec_e, ix.bins_e, ix.use_rec != 0, ix.e_end, ix.n_contigs, c, valid, tu, a, b, out);
}

// Window of a probe below hi as a 32-bit match mask: bit j set <=> row hi-1-j overlaps.  The scan
// stops at the first row whose prefix max fails "q.start (<) pmax".  Four (end,pmax) pairs are
// fetched per round so the dependent-load chain is a quarter of the window length.  Returns false
// when the window is longer than 32 rows (the caller then counts it with the whole wavefront).
template <bool STRICT>
__device__ __forceinline__ bool window_mask(const IndexView& ix, int a, int hi, int32_t qs, uint32_t& mask, int& cnt) {
    mask = 0; cnt = 0;
    const int top = hi - 1;
    int p = top;
    // rows are fetched as 32-byte aligned groups of four (end,pmax) pairs: two 16-byte loads per
    // group, both in one 64-byte line.  Rows of the group above p or below a are ignored (the
    // array is padded, so the loads stay in bounds).
    while (p >= a) {
        const int base = p & ~3;
        const int4 v01 = *reinterpret_cast<const int4*>(ix.ep + base);
        int4 v23 = make_int4(0, 0, 0, 0);                   // rows base+2, base+3: only when p reaches them
        if ((p & 3) >= 2) v23 = *reinterpret_cast<const int4*>(ix.ep + base + 2);
        const int32_t en[4] = {v01.x, v01.z, v23.x, v23.z};
        const int32_t pm[4] = {v01.y, v01.w, v23.y, v23.w};
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            const int idx = base + j;
            if (idx > p) continue;
            if (idx < a || !lt_op<STRICT>(qs, pm[j])) { cnt = __popc(mask); return true; }
            if (top - idx >= 32) { cnt = 0; return false; }   // longer than the mask: counted cooperatively
            if (lt_op<STRICT>(qs, en[j])) mask |= 1u << (top - idx);
        }
        p = base - 1;
    }
    cnt = __popc(mask);
    return true;
}

// Long windows (> 32 rows: dense / deeply nested build sides) are handled by the whole wavefront,
// one probe at a time: lane l looks at row p0 - l, so a step covers 64 consecutive rows with one
// coalesced 512-byte read; "q.start (<) pmax" holds for a prefix of the lanes (pmax is
// non-decreasing in the position), a ballot finds where the window ends and a popcount of the
// match ballot counts it.
template <bool STRICT>
__device__ __forceinline__ int wave_count_window(const IndexView& ix, int a, int hi, int32_t qs) {
    const int lane = threadIdx.x & (kWave - 1);
    int cnt = 0;
    for (int p0 = hi - 1; p0 >= a; p0 -= kWave) {
        const int p = p0 - lane;
        int2 v = make_int2(0, 0);
        if (p >= a) v = ix.ep[p];
        const bool pass = p >= a && lt_op<STRICT>(qs, v.y);
        const bool match = pass && lt_op<STRICT>(qs, v.x);
        cnt += (int)__popcll(__ballot(match));
        if (__popcll(__ballot(pass)) < kWave) break;
    }
    return cnt;
}

// exact count by the bounded backward scan (valid for every input, including
// zero-length and inverted rows)
template <bool STRICT>
__device__ __forceinline__ int scan_count(const IndexView& ix, int a, int hi, int32_t qs) {
    int cnt = 0;
    for (int p = hi - 1; p >= a; --p) {
        const int2 v = ix.ep[p];
        if (!lt_op<STRICT>(qs, v.y)) break;
        cnt += lt_op<STRICT>(qs, v.x) ? 1 : 0;
    }
    return cnt;
}

// Load / store N consecutive int32 of one thread (16- or 8-byte vector access when the group is
// complete and the column is 16-byte aligned; i0 is a multiple of N).
template <int N>
__device__ __forceinline__ void load_items(const int32_t* __restrict__ p, int64_t i0, int64_t n, bool vec_ok,
                                           int32_t fill, int32_t (&out)[N]) {
    if (vec_ok && i0 + N <= n) {
        if constexpr (N == 4) {
            const int4 v = *reinterpret_cast<const int4*>(p + i0);
            out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
            return;
        } else if constexpr (N == 2) {
            const int2 v = *reinterpret_cast<const int2*>(p + i0);
            out[0] = v.x; out[1] = v.y;
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = (i0 + k < n) ? p[i0 + k] : fill;
}
template <int N>
__device__ __forceinline__ void store_items(int32_t* __restrict__ p, int64_t i0, int64_t n, bool vec_ok,
                                            const int32_t (&v)[N]) {
    if (vec_ok && i0 + N <= n) {
        if constexpr (N == 4) { *reinterpret_cast<int4*>(p + i0) = make_int4(v[0], v[1], v[2], v[3]); return; }
        else if constexpr (N == 2) { *reinterpret_cast<int2*>(p + i0) = make_int2(v[0], v[1]); return; }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) if (i0 + k < n) p[i0 + k] = v[k];
}

// ------------------------------------------------------------------ index build

__global__ void k_iota_flip(const int32_t* __restrict__ coord, int64_t n, uint32_t* __restrict__ keys,
                            uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = flip(coord[i]); vals[i] = (uint32_t)i; }
}

// keys[i] = contig id of the row at sorted position i, clamped to n_contigs when outside the dictionary
__global__ void k_gather_contig(const int32_t* __restrict__ contig, const uint32_t* __restrict__ rows, int64_t n,
                                int32_t n_contigs, uint32_t* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int32_t c = contig[rows[i]];
        keys[i] = ((uint32_t)c < (uint32_t)n_contigs) ? (uint32_t)c : (uint32_t)n_contigs;
    }
}

// After the final pass: materialise the sorted columns, the (contig,end) composite for the
// prefix-max scan, the segment offsets and the inverted-row flag.
__global__ void k_index_finalize(const int32_t* __restrict__ start, const int32_t* __restrict__ end,
                                 const uint32_t* __restrict__ rows, const uint32_t* __restrict__ ckeys,
                                 const int32_t* __restrict__ row_id, int64_t n,
                                 int32_t n_contigs, int32_t* __restrict__ b_start, int32_t* __restrict__ b_row,
                                 int32_t* __restrict__ b_contig, unsigned long long* __restrict__ comp,
                                 int32_t* __restrict__ seg, int32_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = rows[i];
    const uint32_t c = ckeys[i];
    const int32_t s = start[r], e = end[r];
    b_start[i] = s;
    b_row[i] = row_id ? row_id[r] : (int32_t)r;
    b_contig[i] = (int32_t)c;
    comp[i] = ((unsigned long long)c << 32) | (unsigned long long)flip(e);
    if (s > e && c < (uint32_t)n_contigs) flags[0] = 1;
    // seg[k] = first position whose contig key is >= k, for k in (prev, c]
    const int32_t prev = (i == 0) ? -1 : (int32_t)ckeys[i - 1];
    for (int32_t k = prev + 1; k <= (int32_t)c; ++k) seg[k] = (int32_t)i;
    if (i == n - 1)
        for (int32_t k = (int32_t)c + 1; k <= n_contigs + 1; ++k) seg[k] = (int32_t)n;
}

__global__ void k_emit_ep(const unsigned long long* __restrict__ comp_raw, const unsigned long long* __restrict__ comp_max,
                          int64_t n, int2* __restrict__ ep) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ep[i] = make_int2(unflip((uint32_t)comp_raw[i]), unflip((uint32_t)comp_max[i]));
}

__global__ void k_end_keys(const int2* __restrict__ ep, int64_t n, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = flip(ep[i].x); vals[i] = (uint32_t)i; }
}
__global__ void k_gather_u32(const int32_t* __restrict__ src, const uint32_t* __restrict__ pos, int64_t n,
                             uint32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)src[pos[i]];
}
__global__ void k_end_finalize(const int2* __restrict__ ep, const uint32_t* __restrict__ pos, int64_t n,
                               int32_t* __restrict__ e_end, int32_t* __restrict__ e_pos) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const uint32_t p = pos[i]; e_end[i] = ep[p].x; e_pos[i] = (int32_t)p; }
}

// change[p] = p where the prefix max changes (or the segment starts), else 0; an inclusive max-scan
// turns it into pargmax[p] = position of the first row attaining the prefix max at p.
__global__ void k_pmax_change(const int2* __restrict__ ep, const int32_t* __restrict__ b_contig, int64_t n,
                              uint32_t* __restrict__ change) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const bool first = p == 0 || b_contig[p] != b_contig[p - 1] || ep[p].y != ep[p - 1].y;
    change[p] = first ? (uint32_t)p : 0u;
}

// count_overlaps: joint bin grid.  Both rank queries of a probe -- #{start (<) q.end} over the
// start order and #{!(q.start (<) end)} over the end order -- use the SAME coordinate bins, and a
// read is ~125 bp long while a bin is thousands of bp wide, so q.start and q.end almost always
// fall into one bin: ONE 16-byte gather answers both ranks.
__global__ void k_contig_meta_joint(const int32_t* __restrict__ seg, const int32_t* __restrict__ b_start,
                                    const int32_t* __restrict__ e_end, int32_t n_contigs, int4* __restrict__ cmeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_contigs) return;
    const int a = seg[c], b = seg[c + 1];
    uint32_t ulo = 0, uhi = 0;
    int shift = 0;
    if (b > a) {
        const uint32_t s0 = flip(b_start[a]), e0 = flip(e_end[a]), s1 = flip(b_start[b - 1]), e1 = flip(e_end[b - 1]);
        ulo = s0 < e0 ? s0 : e0; uhi = s1 > e1 ? s1 : e1;
        const unsigned long long span = (unsigned long long)(uhi - ulo), cap = 2ull * (unsigned long long)(b - a);
        while ((span >> shift) + 1ull > cap) ++shift;
    }
    cmeta[2 * c] = make_int4(a, b, (int)ulo, (int)uhi);
    cmeta[2 * c + 1] = make_int4(shift, 2 * a + 2 * c, 0, 0);
}

__global__ void k_joint_records(const uint32_t* __restrict__ bins_s, const uint32_t* __restrict__ bins_e, int64_t bins_len,
                                const int32_t* __restrict__ b_start, const int32_t* __restrict__ e_end,
                                const int4* __restrict__ cmeta, int32_t n_contigs, int4* __restrict__ crec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bins_len) return;
    int lo = 0, hi = n_contigs;
    while (lo < hi) { const int m = (lo + hi) >> 1; if ((int64_t)cmeta[2 * m + 1].y <= i) lo = m + 1; else hi = m; }
    const int c = lo - 1;
    const int ps = (int)bins_s[i], pe = (int)bins_e[i];
    int32_t ks = 0x7fffffff, ke = 0x7fffffff;
    if (c >= 0) {
        const int bend = cmeta[2 * c].y;
        if (ps < bend) ks = b_start[ps];
        if (pe < bend) ke = e_end[pe];
    }
    crec[i] = make_int4(ps, ks, pe, ke);
}

// nearest (k = 1): everything the no-overlap case needs about a bound position p in ONE 16-byte
// record: the best row on the left (largest end among rows < p: value and build row) and the row
// at p (start, end).  n + 1 records; the fields that do not exist (p = 0 / p = n) are never read.
__global__ void k_nearest_records(const int32_t* __restrict__ b_start, const int2* __restrict__ ep,
                                  const int32_t* __restrict__ b_row, const int32_t* __restrict__ pargmax, int64_t n,
                                  int4* __restrict__ nrec) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p > n) return;
    int4 r = make_int4(0, -1, 0, 0);
    if (p >= 1) { r.x = ep[p - 1].y; r.y = b_row[pargmax[p - 1]]; }
    if (p < n) { r.z = b_start[p]; r.w = ep[p].x; }
    nrec[p] = r;
}

// Per-contig metadata of the direct-address table: bin width 2^shift chosen so that the contig has
// at most 2 n_c bins (about one build row per bin for evenly spread rows); its slice of the table
// starts at tb = 2 a + 2 c.
__global__ void k_contig_meta(const int32_t* __restrict__ seg, const int32_t* __restrict__ b_start, int32_t n_contigs,
                              int4* __restrict__ cmeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_contigs) return;
    const int a = seg[c], b = seg[c + 1];
    uint32_t ulo = 0, uhi = 0;
    int shift = 0;
    if (b > a) {
        ulo = flip(b_start[a]); uhi = flip(b_start[b - 1]);
        const unsigned long long span = (unsigned long long)(uhi - ulo), cap = 2ull * (unsigned long long)(b - a);
        while ((span >> shift) + 1ull > cap) ++shift;
    }
    cmeta[2 * c] = make_int4(a, b, (int)ulo, (int)uhi);
    cmeta[2 * c + 1] = make_int4(shift, 2 * a + 2 * c, 0, 0);
}

// bins (zero-filled) receives, for the last row p of every non-empty bin j, the value p + 1 at slot
// j + 1, and a at slot 0 of every contig; an inclusive max-scan over the whole table then yields
// bins[tb + k] = first position whose start falls in bin >= k (positions grow with the table index).
__global__ void k_bins_mark(const int32_t* __restrict__ b_start, const int32_t* __restrict__ b_contig, int64_t n,
                            int32_t n_contigs, const int4* __restrict__ cmeta, uint32_t* __restrict__ bins) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t c = b_contig[p];
    if ((uint32_t)c >= (uint32_t)n_contigs) return;
    const int4 m0 = cmeta[2 * c], m1 = cmeta[2 * c + 1];
    const uint32_t ulo = (uint32_t)m0.z;
    const uint32_t j = (flip(b_start[p]) - ulo) >> m1.x;
    const bool last = (p == m0.y - 1) || (((flip(b_start[p + 1]) - ulo) >> m1.x) > j);
    if (last) bins[(uint32_t)m1.y + j + 1] = (uint32_t)p + 1u;
    if (p == m0.x) bins[(uint32_t)m1.y] = (uint32_t)p;
}

// brec[i] = {p0, key[p0], key[p0+1], key[p0+2]} for table slot i (p0 = bins[i] after the max-scan);
// keys past the end of the slot's contig segment are replaced by INT32_MAX (compares as "not below"
// any reachable target).  The contig of a slot is found by a bound search over the table offsets.
__global__ void k_bins_records(const uint32_t* __restrict__ bins, int64_t bins_len, const int32_t* __restrict__ keys,
                               const int4* __restrict__ cmeta, int32_t n_contigs, int4* __restrict__ brec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bins_len) return;
    // last contig whose table offset tb = cmeta[2c+1].y is <= i
    int lo = 0, hi = n_contigs;
    while (lo < hi) { const int m = (lo + hi) >> 1; if ((int64_t)cmeta[2 * m + 1].y <= i) lo = m + 1; else hi = m; }
    const int c = lo - 1;
    const int p0 = (int)bins[i];
    int32_t k0 = 0x7fffffff, k1 = 0x7fffffff, k2 = 0x7fffffff;
    if (c >= 0) {
        const int bend = cmeta[2 * c].y;
        if (p0 < bend) k0 = keys[p0];
        if (p0 + 1 < bend) k1 = keys[p0 + 1];
        if (p0 + 2 < bend) k2 = keys[p0 + 2];
    }
    brec[i] = make_int4(p0, k0, k1, k2);
}

// ------------------------------------------------------------------ overlap: count -> fill

// ---- shared bodies of the count / fill / fused kernels -------------------------------------------

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
        // wavefront-cooperative exact count of every long window of this round (uniform loop);
        // four windows per step so that four first-chunk reads are in flight together
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
                int cc = 0;
                int2 v = v0[t];
                int step = 0;
                for (int p0 = chi[t] - 1; p0 >= ca[t]; p0 -= kWave, ++step) {
                    const int p = p0 - lane;
                    if (step == 1) v = v1[t];
                    else if (step > 1) v = (p >= ca[t]) ? ix.ep[p] : make_int2(0, 0);
                    const bool pass = p >= ca[t] && lt_op<STRICT>(cqs[t], v.y);
                    const bool match = pass && lt_op<STRICT>(cqs[t], v.x);
                    cc += (int)__popcll(__ballot(match));
                    if (__popcll(__ballot(pass)) < kWave) break;
                }
                if (lane == src[t]) { cn = cc; x[k] = cc; }
            }
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

struct GlobalRow {
    const int32_t* b_row;
    __device__ __forceinline__ int32_t operator()(int p) const { return b_row[p]; }
};

template <bool STRICT, int THREADS, int STAGE, class RowOf, int N>
__device__ __forceinline__ void emit_tile_rows(const IndexView& ix, const RowOf& rowof, const int32_t (&hi)[N],
                                               const int32_t (&x)[N], const int32_t (&cnt)[N],
                                               const int32_t (&row)[N], const int32_t (&qs)[N],
                                               long long loc0, long long tot, long long tbase, int32_t* st_p, int32_t* st_b,
                                               int32_t* __restrict__ out_probe, int32_t* __restrict__ out_build) {
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
                        st_b[o - w0] = ix.b_row[hi[k] - 1 - j];
                    }
                    ++o;
                }
            }
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
                int found = 0;
                for (int p0 = h - 1; found < c && p0 >= 0 && cend - found > w0; p0 -= kWave) {
                    const int p = p0 - lane;
                    int2 v = make_int2(0, 0);
                    int32_t br = 0;
                    if (p >= 0) { v = ix.ep[p]; br = ix.b_row[p]; }
                    const bool m = p >= 0 && lt_op<STRICT>(cqs, v.x);
                    const unsigned long long mm = __ballot(m);
                    if (m) {
                        // rows below the window (or of the previous contig) rank past the c-th match
                        const long long o = cend - 1 - found - (long long)__popcll(mm & lt_lanes);
                        if (o >= w0 && o < w1 && o >= cend - c) { st_p[o - w0] = crow; st_b[o - w0] = br; }
                    }
                    found += (int)__popcll(mm);
                }
            }
            off = end;
        }
        __syncthreads();
        const int t = (int)((tot - w0) < (long long)STAGE ? (tot - w0) : (long long)STAGE);
        for (int i = threadIdx.x; i < t; i += THREADS) {
            out_probe[tbase + w0 + i] = st_p[i];
            out_build[tbase + w0 + i] = st_b[i];
        }
        __syncthreads();
    }
}

template <bool STRICT>
__device__ __forceinline__ void emit_tile(const IndexView& ix, const int32_t (&hi)[PROBE_ITEMS],
                                          const int32_t (&x)[PROBE_ITEMS], const int32_t (&cnt)[PROBE_ITEMS],
                                          const int32_t (&row)[PROBE_ITEMS], const int32_t (&qs)[PROBE_ITEMS],
                                          long long loc0, long long tot, long long tbase, int32_t* st_p, int32_t* st_b,
                                          int32_t* __restrict__ out_probe, int32_t* __restrict__ out_build) {
    emit_tile_rows<STRICT, PROBE_THREADS, FILL_STAGE>(ix, GlobalRow{ix.b_row}, hi, x, cnt, row, qs, loc0, tot, tbase, st_p, st_b,
                                                       out_probe, out_build);
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
    const int64_t i0 = (int64_t)blockIdx.x * PROBE_TILE + (int64_t)threadIdx.x * PROBE_ITEMS;
    int32_t c[PROBE_ITEMS], s[PROBE_ITEMS], e[PROBE_ITEMS];
    load_items(pc, i0, n, vec_ok, -1, c);
    load_items(ps, i0, n, vec_ok, 0, s);
    load_items(pe, i0, n, vec_ok, 0, e);
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
    if (threadIdx.x == 0) tile_tot[blockIdx.x] = tot;
}

// Pass 2.  tile_base = exclusive scan of tile_tot.
template <bool STRICT>
__global__ __launch_bounds__(PROBE_THREADS, 6) void k_overlap_fill(IndexView ix, const int32_t* __restrict__ ps, int64_t n,
                                                                bool vec_ok, const int32_t* __restrict__ hi_in,
                                                                const int32_t* __restrict__ cnt_in,
                                                                const long long* __restrict__ tile_base,
                                                                const int32_t* __restrict__ probe_ids,
                                                                int32_t* __restrict__ out_probe,
                                                                int32_t* __restrict__ out_build) {
    __shared__ long long lds[PROBE_THREADS / kWave];
    __shared__ int32_t st_p[FILL_STAGE];
    __shared__ int32_t st_b[FILL_STAGE];
    const int64_t i0 = (int64_t)blockIdx.x * PROBE_TILE + (int64_t)threadIdx.x * PROBE_ITEMS;
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
    emit_tile<STRICT>(ix, hi, x, cnt, row, qs, loc0, tot, tile_base[blockIdx.x], st_p, st_b, out_probe, out_build);
}

// Fused single pass (count + fill) for callers that bring an output buffer of known capacity
// (steady-state / streaming use: the previous batch sized it).  Each tile reserves its output range
// with ONE 64-bit atomicAdd on a cursor, so no tile waits for another and nothing is written to or
// re-read from HBM between counting and emitting.  Tile ranges land in reservation order: the
// pairs of one probe row stay contiguous and ordered, the order of tiles is not reproducible from
// run to run (the two-pass path is the deterministic one).  state[0] = cursor (= total on exit),
// state[1] = 1 when the capacity was exceeded (nothing is written past it).
template <bool STRICT>
__global__ __launch_bounds__(PROBE_THREADS, 6) void k_overlap_fused(IndexView ix, const int32_t* __restrict__ pc,
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
    const int64_t i0 = (int64_t)blockIdx.x * PROBE_TILE + (int64_t)threadIdx.x * PROBE_ITEMS;
    int32_t c[PROBE_ITEMS], s[PROBE_ITEMS], e[PROBE_ITEMS];
    load_items(pc, i0, n, vec_ok, -1, c);
    load_items(ps, i0, n, vec_ok, 0, s);
    load_items(pe, i0, n, vec_ok, 0, e);
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
            out_probe[tbase + w0 + i] = st_p[i];
            out_build[tbase + w0 + i] = st_b[i];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ count_overlaps

// count = #{b.start (<) q.end} - #{!(q.start (<) b.end)}  (two-rank formula of the reference's
// SQL sweep, polars_bio/range_op.py:548-595); the bounded scan replaces it for rows where the
// formula is not exact (zero-length/inverted probe, or any inverted build row).
// rank of a target inside one joint-grid slot: p0/k0 come from the record; when the first row of the
// bin is still below the target look at the next row, and only then bound-search up to the next bin
__device__ __forceinline__ int joint_rank(const int32_t* __restrict__ keys, int p0, int32_t k0, unsigned long long t, int b,
                                          const int4* __restrict__ crec, uint32_t slot, bool end_table) {
    if (!((unsigned long long)flip(k0) < t)) return p0;
    int lo = p0 + 1;
    if (lo < b && (unsigned long long)flip(keys[lo]) < t) {
        ++lo;
        const int4 nx = crec[slot + 1];
        int hi = end_table ? nx.z : nx.x;
        while (lo < hi) {
            const int m = lo + ((hi - lo) >> 1);
            if ((unsigned long long)flip(keys[m]) < t) lo = m + 1; else hi = m;
        }
    }
    return lo;
}

template <bool STRICT, int N>
__global__ __launch_bounds__(PROBE_THREADS) void k_count_overlaps(IndexView ix, const int32_t* __restrict__ pc,
                                                                  const int32_t* __restrict__ ps,
                                                                  const int32_t* __restrict__ pe, int64_t n, bool vec_ok,
                                                                  long long* __restrict__ counts) {
    const int64_t i0 = (int64_t)blockIdx.x * (PROBE_THREADS * N) + (int64_t)threadIdx.x * N;
    int32_t c[N], s[N], e[N];
    load_items(pc, i0, n, vec_ok, -1, c);
    load_items(ps, i0, n, vec_ok, 0, s);
    load_items(pe, i0, n, vec_ok, 0, e);
    const bool inv = ix.flags[0] != 0;
    // phase 1: metadata and the (usually single) record gather of every probe, issued together
    int a[N], b[N];
    unsigned long long te[N], ts[N];
    uint32_t se[N], ss[N];
    int he[N], hs[N];          // 0: rank = a, 1: rank = b, 2: table
    int4 re[N], rs[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const bool ok = i0 + k < n && (uint32_t)c[k] < (uint32_t)ix.n_contigs;
        int4 m0 = make_int4(0, 0, 0, 0), m1 = make_int4(0, 0, 0, 0);
        if (ok) { m0 = ix.cmeta_j[2 * c[k]]; m1 = ix.cmeta_j[2 * c[k] + 1]; }
        a[k] = m0.x; b[k] = m0.y;
        const uint32_t ulo = (uint32_t)m0.z, uhi = (uint32_t)m0.w;
        te[k] = (unsigned long long)flip(e[k]) + (STRICT ? 0ull : 1ull);   // first start >= / > q.end
        ts[k] = (unsigned long long)flip(s[k]) + (STRICT ? 1ull : 0ull);   // first end > / >= q.start
        he[k] = (b[k] <= a[k] || te[k] <= ulo) ? 0 : (te[k] > uhi ? 1 : 2);
        hs[k] = (b[k] <= a[k] || ts[k] <= ulo) ? 0 : (ts[k] > uhi ? 1 : 2);
        se[k] = he[k] == 2 ? (uint32_t)m1.y + (((uint32_t)te[k] - ulo) >> m1.x) : 0u;
        ss[k] = hs[k] == 2 ? (uint32_t)m1.y + (((uint32_t)ts[k] - ulo) >> m1.x) : 0u;
        re[k] = make_int4(0, 0, 0, 0); rs[k] = make_int4(0, 0, 0, 0);
        if (he[k] == 2) re[k] = ix.crec[se[k]];
        if (hs[k] == 2) rs[k] = (he[k] == 2 && ss[k] == se[k]) ? re[k] : ix.crec[ss[k]];
    }
    long long cnt[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int hi = he[k] == 0 ? a[k] : (he[k] == 1 ? b[k] : joint_rank(ix.b_start, re[k].x, re[k].y, te[k], b[k], ix.crec, se[k], false));
        const bool degenerate = inv || (STRICT ? (s[k] >= e[k]) : (s[k] > e[k]));
        if (!degenerate) {
            const int r = hs[k] == 0 ? a[k] : (hs[k] == 1 ? b[k] : joint_rank(ix.e_end, rs[k].z, rs[k].w, ts[k], b[k], ix.crec, ss[k], true));
            cnt[k] = (long long)hi - (long long)r;
        } else {
            cnt[k] = scan_count<STRICT>(ix, a[k], hi, s[k]);
        }
    }
    if (i0 + N <= n && (reinterpret_cast<uintptr_t>(counts) & 15u) == 0 && (N % 2) == 0) {
#pragma unroll
        for (int k = 0; k < N; k += 2)
            reinterpret_cast<longlong2*>(counts + i0)[k / 2] = make_longlong2(cnt[k], cnt[k + 1]);
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) if (i0 + k < n) counts[i0 + k] = cnt[k];
    }
}

// ------------------------------------------------------------------ nearest

// k = 1, include_overlaps = 1 (the default pb.nearest).  An overlapping row wins with distance 0
// (the one with the smallest (start,row): tests/_expected.py:130-172 tie-break); otherwise the
// closer of the row with the largest end before the probe (ties: smallest (start,row)) and the
// row with the smallest start after it; equal distance -> the left one.
template <bool STRICT, int N>
__global__ __launch_bounds__(PROBE_THREADS) void k_nearest_k1(IndexView ix, const int32_t* __restrict__ pc,
                                                              const int32_t* __restrict__ ps,
                                                              const int32_t* __restrict__ pe, int64_t n, bool vec_ok,
                                                              int32_t* __restrict__ out_idx, long long* __restrict__ out_dist,
                                                              int32_t* __restrict__ out_n) {
    const int64_t i0 = (int64_t)blockIdx.x * (PROBE_THREADS * N) + (int64_t)threadIdx.x * N;
    int32_t c[N], s[N], e[N];
    load_items(pc, i0, n, vec_ok, -1, c);
    load_items(ps, i0, n, vec_ok, 0, s);
    load_items(pe, i0, n, vec_ok, 0, e);
    int a[N], b[N], hi[N];
    bool valid[N];
#pragma unroll
    for (int k = 0; k < N; ++k) valid[k] = i0 + k < n;
    bound_hi_tab4<STRICT>(ix, c, valid, e, a, b, hi);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (i0 + k >= n) continue;
        int32_t idx = -1; long long dist = -1; int32_t found = 0;
        if (b[k] > a[k]) {
            const int4 R = ix.nrec[hi[k]];     // {pmax[hi-1], its build row, start[hi], end[hi]}
            const bool have_l = hi[k] > a[k], have_r = hi[k] < b[k];
            if (have_l && lt_op<STRICT>(s[k], R.x)) {
                // some row below hi overlaps: the first position whose prefix max satisfies
                // "q.start (<) pmax" is the overlapping row with the smallest (start,row); walk down
                // from hi-1 while it holds (pmax is non-decreasing), at most 8 rows, then bound-search
                int lo = hi[k] - 1;
                int p = hi[k] - 2, steps = 1;
                while (p >= a[k] && steps < 8 && lt_op<STRICT>(s[k], ix.ep[p].y)) { lo = p; --p; ++steps; }
                if (steps == 8 && p >= a[k]) lo = bound_lo<STRICT>(ix, a[k], p + 1, s[k]);
                idx = ix.b_row[lo]; dist = 0; found = 1;
            } else {
                const long long dl = (long long)s[k] - (long long)R.x;
                const long long dr = have_r ? gap_dist(s[k], e[k], R.z, R.w) : 0;
                if (have_l && (!have_r || dl <= dr)) { idx = R.y; dist = dl; found = 1; }
                else if (have_r) { idx = ix.b_row[hi[k]]; dist = dr; found = 1; }
            }
        }
        out_idx[i0 + k] = idx; out_dist[i0 + k] = dist; out_n[i0 + k] = found;
    }
}

// General k / include_overlaps: per-probe merge of three ordered streams (overlapping rows in
// (start,row) order; "left" rows by end descending; "right" rows by start ascending).
// One thread per probe; k slots per probe, unused slots -1.
template <bool STRICT>
__global__ __launch_bounds__(PROBE_THREADS) void k_nearest_general(IndexView ix, const int32_t* __restrict__ pc,
                                                                   const int32_t* __restrict__ ps,
                                                                   const int32_t* __restrict__ pe, int64_t n, int kk,
                                                                   int include_overlaps, int32_t* __restrict__ out_idx,
                                                                   long long* __restrict__ out_dist,
                                                                   int32_t* __restrict__ out_n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t qs = ps[i], qe = pe[i];
    int32_t* oi = out_idx + i * kk;
    long long* od = out_dist + i * kk;
    for (int r = 0; r < kk; ++r) { oi[r] = -1; od[r] = -1; }
    int a, b;
    seg_bounds(ix, pc[i], true, a, b);
    int found = 0;
    if (b > a) {
        const int hi = bound_hi<STRICT>(ix, a, b, qe);
        if (include_overlaps) {
            const int lo = bound_lo<STRICT>(ix, a, hi, qs);
            for (int p = lo; p < hi && found < kk; ++p)
                if (lt_op<STRICT>(qs, ix.ep[p].x)) { oi[found] = ix.b_row[p]; od[found] = 0; ++found; }
        }
        const int r_top = bound_r<STRICT>(ix, a, b, qs);
        int run_hi = r_top, run_lo = r_top, lp = r_top, rp = hi;
        while (found < kk) {
            for (;;) {
                while (lp < run_hi && ix.e_pos[lp] >= hi) ++lp;   // not class "left": start fails (<) q.end
                if (lp < run_hi || run_lo <= a) break;
                run_hi = run_lo;
                run_lo = bsearch32<false>(ix.e_end, a, run_hi, ix.e_end[run_hi - 1]);
                lp = run_lo;
            }
            const bool have_l = lp < run_hi, have_r = rp < b;
            if (!have_l && !have_r) break;
            long long dl = 0, dr = 0; int pl = 0;
            if (have_l) { pl = ix.e_pos[lp]; dl = gap_dist(qs, qe, ix.b_start[pl], ix.ep[pl].x); }
            if (have_r) dr = gap_dist(qs, qe, ix.b_start[rp], ix.ep[rp].x);
            if (have_l && (!have_r || dl <= dr)) { oi[found] = ix.b_row[pl]; od[found] = dl; ++found; ++lp; }
            else { oi[found] = ix.b_row[rp]; od[found] = dr; ++found; ++rp; }
        }
    }
    out_n[i] = found;
}

}  // namespace ivj

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
    const uint32_t bkt = ((uint32_t)m1.y + j) >> bshift;
    return bkt < (uint32_t)(PART_BUCKETS - 2) ? bkt : (uint32_t)(PART_BUCKETS - 2);
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
    const uint32_t bkt = ((uint32_t)m1.y + j) >> bshift;
    return bkt < (uint32_t)(PART_BUCKETS - 2) ? bkt : (uint32_t)(PART_BUCKETS - 2);
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
        load_items(pc, i0, n, vec_ok, -1, c);
        load_items(pe, i0, n, vec_ok, 0, e);
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
    IVJ_PART_EXCHANGE(r, orow);
#undef IVJ_PART_EXCHANGE
}

}  // namespace ivj
