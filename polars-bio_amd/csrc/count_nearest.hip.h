// count_nearest.hip.h -- pb.count_overlaps (joint bin grid) and pb.nearest (k = 1 records, general k merge) kernels.
#pragma once
#include "index_view.hip.h"

namespace ivj {

// ------------------------------------------------------------------ count_overlaps

// count = #{b.start (<) q.end} - #{!(q.start (<) b.end)}  (two-rank formula of the reference's
// SQL sweep, polars_bio/range_op.py:548-595); the bounded scan replaces it for rows where the
// formula is not exact (zero-length/inverted probe, or any inverted build row).
// Rank of a target t inside its joint-grid bin from the bin's record word pw = first position | more << 31 and its two
// inline key offsets (index_build.hip.h, k_joint_records): toff = t - lower edge of the bin.  Only a bin with a third row
// whose second row is still below the target (or any non-empty bin of a grid wider than 2^16 per bin) touches the key
// array: gallop up from the rows already counted, bounded by the contig segment, then a bound search.
// The kernel's speed is set by the number of L2 requests in flight per CU (the vector L1 holds ~90 outstanding, PMC:
// profiles/r02/pmc_sq_tcp_count_200M_200k.json), so every key-array touch that the record answers instead is time.
__device__ __forceinline__ int joint_rank(const int32_t* __restrict__ keys, int pw, uint32_t offs, uint32_t toff, bool wide,
                                          unsigned long long t, int seg_end) {
    int lo = pw & 0x7fffffff;
    if (!wide) {
        const bool b0 = (offs & 0xffffu) < toff, b1 = (offs >> 16) < toff;
        lo += (b0 ? 1 : 0) + (b1 ? 1 : 0);
        if (!(pw < 0 && b1)) return lo;
    } else if (pw >= 0) return lo;
    int step = 1;
    while (lo + step - 1 < seg_end && (unsigned long long)flip(keys[lo + step - 1]) < t) { lo += step; step <<= 1; }
    int hi = lo + step - 1 < seg_end ? lo + step - 1 : seg_end;
    while (lo < hi) {
        const int m = lo + ((hi - lo) >> 1);
        if ((unsigned long long)flip(keys[m]) < t) lo = m + 1; else hi = m;
    }
    return lo;
}

// LM: the per-contig grid metadata is copied to LDS once per workgroup (n_contigs <= CM_LDS).  Read from global memory the
// two 16-byte metadata loads of a probe are L1 hits, but with a random contig per lane every lane is its own L1 access:
// they were HALF of the kernel's 4.1 L1 accesses per probe (profiles/r01/v17_pmc_sq_tcp_count_200M_200k.json).
constexpr int COUNT_TILES_PER_WG = 4;

template <bool STRICT, int N, bool LM>
__global__ __launch_bounds__(PROBE_THREADS) void k_count_overlaps(IndexView ix, const int32_t* __restrict__ pc,
                                                                  const int32_t* __restrict__ ps,
                                                                  const int32_t* __restrict__ pe, int64_t n, bool vec_ok,
                                                                  long long* __restrict__ counts, int32_t* __restrict__ counts32, int ablate) {
    // counts32 != nullptr: the counts as int32 (a count is bounded by the build rows) into counts32 instead of int64 into counts -- the
    // wire format of the per-probe exchange (host_comm.hip.h), written here so that no pack pass re-reads 8 bytes per probe
    __shared__ int4 l_cm[LM ? 2 * CM_LDS : 1];
    if (LM) {
        for (int i = threadIdx.x; i < 2 * ix.n_contigs; i += PROBE_THREADS) l_cm[i] = ix.cmeta_j[i];
        __syncthreads();
    }
    const bool inv = ix.flags[0] != 0;
#pragma unroll 1
  for (int t = 0; t < COUNT_TILES_PER_WG; ++t) {
    const int64_t i0 = ((int64_t)blockIdx.x * COUNT_TILES_PER_WG + t) * (PROBE_THREADS * N) + (int64_t)threadIdx.x * N;
    if (i0 - (int64_t)threadIdx.x * N >= n) break;
    int32_t c[N], s[N], e[N];
    load_items_nt(pc, i0, n, vec_ok, -1, c);
    load_items_nt(ps, i0, n, vec_ok, 0, s);
    load_items_nt(pe, i0, n, vec_ok, 0, e);
    // phase 1: metadata and the (usually single) record gather of every probe, issued together
    int a[N], b[N];
    unsigned long long te[N], ts[N];
    int he[N], hs[N];          // 0: rank = a, 1: rank = b, 2: table
    bool same[N], wide[N];
    uint32_t oe[N], os[N];     // offset of the target inside its bin
    int4 re[N], rs[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const bool ok = i0 + k < n && (uint32_t)c[k] < (uint32_t)ix.n_contigs;
        int4 m0 = make_int4(0, 0, 0, 0), m1 = make_int4(0, 0, 0, 0);
        if (ok) {
            if (LM) { m0 = l_cm[2 * c[k]]; m1 = l_cm[2 * c[k] + 1]; }
            else { m0 = ix.cmeta_j[2 * c[k]]; m1 = ix.cmeta_j[2 * c[k] + 1]; }
        }
        a[k] = m0.x; b[k] = m0.y;
        const uint32_t ulo = (uint32_t)m0.z, uhi = (uint32_t)m0.w;
        te[k] = (unsigned long long)flip(e[k]) + (STRICT ? 0ull : 1ull);   // first start >= / > q.end
        ts[k] = (unsigned long long)flip(s[k]) + (STRICT ? 1ull : 0ull);   // first end > / >= q.start
        he[k] = (b[k] <= a[k] || te[k] <= ulo) ? 0 : (te[k] > uhi ? 1 : 2);
        hs[k] = (b[k] <= a[k] || ts[k] <= ulo) ? 0 : (ts[k] > uhi ? 1 : 2);
        if (ablate & 1) { he[k] = 0; hs[k] = 0; }
        const uint32_t de = (uint32_t)te[k] - ulo, ds = (uint32_t)ts[k] - ulo, bmask = (1u << m1.x) - 1u;   // shift <= 31
        const uint32_t se = he[k] == 2 ? (uint32_t)m1.y + (de >> m1.x) : 0u;
        const uint32_t ss = hs[k] == 2 ? (uint32_t)m1.y + (ds >> m1.x) : 0u;
        oe[k] = de & bmask; os[k] = ds & bmask; wide[k] = m1.x > 16;
        same[k] = he[k] == 2 && hs[k] == 2 && se == ss;                   // a probe is short against a bin: the usual case
        re[k] = make_int4(0, 0, 0, 0); rs[k] = make_int4(0, 0, 0, 0);
        if (he[k] == 2) re[k] = ix.crec[se];
        if (hs[k] == 2 && !same[k]) rs[k] = ix.crec[ss];
    }
    long long cnt[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (same[k]) rs[k] = re[k];
        const int hi = he[k] == 0 ? a[k] : (he[k] == 1 ? b[k] : joint_rank(ix.b_start, re[k].x, (uint32_t)re[k].z, oe[k], wide[k], te[k], b[k]));
        const bool degenerate = inv || (STRICT ? (s[k] >= e[k]) : (s[k] > e[k]));
        if (!degenerate) {
            const int r = hs[k] == 0 ? a[k] : (hs[k] == 1 ? b[k] : joint_rank(ix.e_end, rs[k].y, (uint32_t)rs[k].w, os[k], wide[k], ts[k], b[k]));
            cnt[k] = (long long)hi - (long long)r;
        } else {
            cnt[k] = scan_count<STRICT>(ix, a[k], hi, s[k]);
        }
    }
    if ((ablate & 2) && cnt[0] != 123456789) continue;
    if (counts32) {
        if (N == 2 && i0 + N <= n && (reinterpret_cast<uintptr_t>(counts32) & 7u) == 0) {
            typedef int v2i __attribute__((ext_vector_type(2)));
            v2i v; v.x = (int)cnt[0]; v.y = (int)cnt[N - 1];
            __builtin_nontemporal_store(v, reinterpret_cast<v2i*>(counts32 + i0));
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) if (i0 + k < n) counts32[i0 + k] = (int32_t)cnt[k];
        }
    } else if (i0 + N <= n && (reinterpret_cast<uintptr_t>(counts) & 15u) == 0 && (N % 2) == 0) {
#pragma unroll
        for (int k = 0; k < N; k += 2)
        {
            typedef long long v2ll __attribute__((ext_vector_type(2)));
            v2ll v; v.x = cnt[k]; v.y = cnt[k + 1];
            __builtin_nontemporal_store(v, reinterpret_cast<v2ll*>(counts + i0) + k / 2);
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) if (i0 + k < n) counts[i0 + k] = cnt[k];
    }
  }
}

// ------------------------------------------------------------------ nearest

__device__ __forceinline__ int4 sel3(int m0, const int4& a, int m1, const int4& b, int m2, const int4& c) {
    return make_int4((a.x & m0) | (b.x & m1) | (c.x & m2), (a.y & m0) | (b.y & m1) | (c.y & m2), (a.z & m0) | (b.z & m1) | (c.z & m2),
                     (a.w & m0) | (b.w & m1) | (c.w & m2));
}

// bound_lo for a probe that is known to overlap a row below hi: the first position whose prefix max satisfies "q.start (<) pmax" lies a few
// rows below hi, so gallop down from hi (2, 8, 32, ... rows) to a position that fails and search the rows above it: ~ 5 dependent
// reads instead of log2(rows of the contig) -- what a probe pays when the index is not L2-resident.
template <bool STRICT>
__device__ __forceinline__ int bound_lo_near(const IndexView& ix, int a, int hi, int32_t qs) {
    int lo = a, step = 2;
    for (;;) {
        const int p = hi - step;
        if (p <= a) break;
        if (!lt_op<STRICT>(qs, ix.ep[p].y)) { lo = p + 1; break; }
        step <<= 2;
    }
    return bound_lo<STRICT>(ix, lo, hi, qs);
}

// The answer of one probe from the record of its hi-bound (R = nrec[2 hi], Q = nrec[2 hi + 1]); [a, b) = the contig's segment, b > a.
template <bool STRICT>
__device__ __forceinline__ void nearest_k1_resolve(const IndexView& ix, int a, int b, int hi, int32_t s, int32_t e, const int4& R, const int4& Q,
                                                   int32_t& idx, long long& dist, int32_t& found) {
    const bool have_l = hi > a, have_r = hi < b;
    if (have_l && lt_op<STRICT>(s, R.x)) {
        // some row below hi overlaps.  The overlapping row with the smallest (start,row) is the first position whose
        // prefix max satisfies "q.start (<) pmax", i.e. the first row of the earliest prefix-max level above q.start:
        // level m (R), m-1 (Q) come with their build rows; a probe below level m-2 as well takes the bound search
        if (!(Q.z >= 0 && lt_op<STRICT>(s, Q.y))) idx = R.y;
        else if (!lt_op<STRICT>(s, Q.w)) idx = Q.z;
        else idx = ix.b_row[bound_lo_near<STRICT>(ix, a, hi, s)];
        dist = 0; found = 1;
    } else {
        const long long dl = (long long)s - (long long)R.x;
        const long long dr = have_r ? gap_dist(s, e, R.z, R.w) : 0;
        if (have_l && (!have_r || dl <= dr)) { idx = R.y; dist = dl; found = 1; }
        else if (have_r) { idx = Q.x; dist = dr; found = 1; }
    }
}

// k = 1, include_overlaps = 1 (the default pb.nearest).  An overlapping row wins with distance 0
// (the one with the smallest (start,row): tests/_expected.py:130-172 tie-break); otherwise the
// closer of the row with the largest end before the probe (ties: smallest (start,row)) and the
// row with the smallest start after it; equal distance -> the left one.
template <bool STRICT, int N>
__global__ __launch_bounds__(PROBE_THREADS) void k_nearest_k1(IndexView ix, const int32_t* __restrict__ pc,
                                                              const int32_t* __restrict__ ps,
                                                              const int32_t* __restrict__ pe, int64_t n, bool vec_ok,
                                                              const int32_t* __restrict__ out_row,
                                                              int32_t* __restrict__ out_idx, long long* __restrict__ out_dist,
                                                              int32_t* __restrict__ out_n, int ablate) {
    // out_row: the probes are a bucketed permutation (partition.hip.h); results go to the original rows
    const long long ntiles = (n + PROBE_THREADS * N - 1) / (PROBE_THREADS * N);
    const long long tile = xcd_tile64(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int64_t i0 = (int64_t)tile * (PROBE_THREADS * N) + (int64_t)threadIdx.x * N;
    int32_t c[N], s[N], e[N];
    load_items(pc, i0, n, vec_ok, -1, c);
    load_items(ps, i0, n, vec_ok, 0, s);
    load_items(pe, i0, n, vec_ok, 0, e);
    int a[N], b[N], hi[N];
    bool valid[N];
#pragma unroll
    for (int k = 0; k < N; ++k) valid[k] = i0 + k < n;
    if (ablate & 1) {
#pragma unroll
        for (int k = 0; k < N; ++k) { a[k] = 0; b[k] = valid[k] ? 1 : 0; hi[k] = (s[k] >> 8) & 0xffff; }      // profiling: no table lookup
    } else bound_hi_tab4<STRICT>(ix, c, valid, e, a, b, hi);
    // ONE 32-byte record per probe (both halves requested together): no walk down the prefix max, no separate build-row gather
    int4 R[N], Q[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        R[k] = make_int4(0, -1, 0, 0); Q[k] = make_int4(-1, 0, -1, (int)0x80000000);
        if (i0 + k < n && b[k] > a[k] && !(ablate & 2)) { R[k] = ix.nrec[2 * (int64_t)hi[k]]; Q[k] = ix.nrec[2 * (int64_t)hi[k] + 1]; }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (i0 + k >= n) continue;
        int32_t idx = -1; long long dist = -1; int32_t found = 0;
        if (b[k] > a[k]) nearest_k1_resolve<STRICT>(ix, a[k], b[k], hi[k], s[k], e[k], R[k], Q[k], idx, dist, found);
        const int64_t o = out_row ? (int64_t)out_row[i0 + k] : i0 + k;
        if ((ablate & 4) && idx != 123456789) continue;
        out_idx[o] = idx; out_dist[o] = dist; out_n[o] = found;
    }
}

// k = 1 over the nearest LINES (index_build.hip.h, k_nearest_lines): probes in INPUT order, no bucketing, no inverse permutation.  One
// 64-byte line per probe (round 5; round 4: 128 bytes) -- the nearest record of the first position of the bin its end falls into AND
// the three rows from there on, from which the records of the positions hi can take are replayed in registers -- fetched with four
// independent 16-byte loads; only a bin with a fourth row below the probe's end, or a probe outside its contig's table, takes the
// second (dependent) gather from nrec.  Per-contig metadata in LDS (n_contigs <= CM_LDS).
template <bool STRICT, int N>
__global__ __launch_bounds__(PROBE_THREADS, 8) void k_nearest_k1_lines(IndexView ix, const int32_t* __restrict__ pc, const int32_t* __restrict__ ps,
                                                                    const int32_t* __restrict__ pe, int64_t n, bool vec_ok,
                                                                    int32_t* __restrict__ out_idx, long long* __restrict__ out_dist,
                                                                    int32_t* __restrict__ out_n, unsigned long long* __restrict__ rest, int64_t n_words) {
    __shared__ int4 l_cm[2 * CM_LDS];
    for (int i = threadIdx.x; i < 2 * ix.n_contigs; i += PROBE_THREADS) l_cm[i] = ix.cmeta[i];
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * (PROBE_THREADS * N) + (int64_t)threadIdx.x * N;
    if (i0 >= n) {
        // a wavefront that lies wholly beyond the probes still owns N mask words: k_nearest_k1_rest reads every word of the last
        // tile, and the scratch they live in is not cleared (lane 0 holds the wavefront's smallest row)
        if ((threadIdx.x & (kWave - 1)) == 0) {
            const int64_t wf0 = ((int64_t)blockIdx.x * PROBE_THREADS + threadIdx.x) / kWave;
#pragma unroll
            for (int k = 0; k < N; ++k) { rest[N * wf0 + k] = 0ull; rest[n_words + N * wf0 + k] = 0ull; }
        }
        return;
    }
    int32_t c[N], s[N], e[N];
    load_items_nt(pc, i0, n, vec_ok, -1, c);
    load_items_nt(ps, i0, n, vec_ok, 0, s);
    load_items_nt(pe, i0, n, vec_ok, 0, e);
    int a[N], b[N], hi[N];
    unsigned long long tu[N];
    bool inb[N];
    int4 W0[N], W1[N], W2[N], W3[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        W0[k] = W1[k] = W2[k] = W3[k] = make_int4(0, 0, 0, 0);
        const bool ok = i0 + k < n && (uint32_t)c[k] < (uint32_t)ix.n_contigs;
        int4 m0 = make_int4(0, 0, 0, 0), m1 = make_int4(0, 0, 0, 0);
        if (ok) { m0 = l_cm[2 * c[k]]; m1 = l_cm[2 * c[k] + 1]; }
        a[k] = m0.x; b[k] = m0.y;
        const uint32_t ulo = (uint32_t)m0.z;
        tu[k] = (unsigned long long)flip(e[k]) + (STRICT ? 0ull : 1ull);
        inb[k] = false; hi[k] = a[k];
        if (b[k] <= a[k] || tu[k] <= (unsigned long long)ulo) hi[k] = a[k];
        else if (tu[k] > (unsigned long long)(uint32_t)m0.w) hi[k] = b[k];
        else {
            inb[k] = true;
            const int4* line = ix.nline + 4 * (int64_t)((uint32_t)m1.y + (((uint32_t)tu[k] - ulo) >> m1.x));
            W0[k] = line[0]; W1[k] = line[1]; W2[k] = line[2]; W3[k] = line[3];
        }
    }
    int4 R[N], Q[N];
    bool far[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        R[k] = make_int4(0, -1, 0, 0); Q[k] = make_int4(-1, 0, -1, (int)0x80000000);
        far[k] = i0 + k < n && b[k] > a[k];                                   // the record comes from nrec (outside the table / a fourth row of the bin)
        if (inb[k]) {
            // the 64-byte line (index_build.hip.h, k_nearest_lines): {p0 | first << 30, record of p0, rows p0, p0 + 1, p0 + 2}
            const int p0 = W0[k].x & 0x3fffffff;
            const bool first = (W0[k].x & 0x40000000) != 0;
            int32_t pm = W0[k].y, ar = W0[k].z, l1v = W0[k].w, l1r = W1[k].x, l2v = W1[k].y;
            const int32_t s0 = W1[k].z, e0 = W1[k].w, r0 = W2[k].x, s1 = W2[k].y, e1 = W2[k].z, r1 = W2[k].w, s2 = W3[k].x, e2 = W3[k].y, r2 = W3[k].z;
            // rank inside the bin: rows of later bins start above the target, rows past the segment carry INT32_MAX
            // (a row past the segment has build row -1: it must not count even for the one target above INT32_MAX, a Weak probe ending there)
            const bool n0 = r0 >= 0 && (unsigned long long)flip(s0) < tu[k];
            const bool n1 = n0 && r1 >= 0 && (unsigned long long)flip(s1) < tu[k];
            const bool n2 = n1 && r2 >= 0 && (unsigned long long)flip(s2) < tu[k];
            const int d = (n0 ? 1 : 0) + (n1 ? 1 : 0) + (n2 ? 1 : 0);
            hi[k] = p0 + d;
            if (d <= 2) {
                far[k] = false;
                // the record of p + 1 from the record of p and row p: a row that ends above the prefix max so far opens a new level on
                // top (it becomes the first row attaining the maximum), the old levels move one down; otherwise nothing changes.  The
                // first row of a segment has nothing below it.  (Written with selects: no divergent control flow, no indexed registers.)
                bool top_first = first;                                        // "the position being pushed is its segment's first row"
                if (d >= 1) {
                    const bool up = top_first || e0 > pm;
                    const int32_t o_pm = pm, o_ar = ar, o_l1v = l1v, o_l1r = l1r;
                    l2v = up ? (top_first || o_l1r < 0 ? (int32_t)0x80000000 : o_l1v) : l2v;
                    l1v = up ? (top_first ? 0 : o_pm) : l1v;
                    l1r = up ? (top_first ? -1 : o_ar) : l1r;
                    pm = up ? e0 : pm;
                    ar = up ? r0 : ar;
                    top_first = false;
                }
                if (d >= 2) {
                    const bool up = e1 > pm;
                    const int32_t o_pm = pm, o_ar = ar, o_l1v = l1v, o_l1r = l1r;
                    l2v = up ? (o_l1r < 0 ? (int32_t)0x80000000 : o_l1v) : l2v;
                    l1v = up ? o_pm : l1v;
                    l1r = up ? o_ar : l1r;
                    pm = up ? e1 : pm;
                    ar = up ? r1 : ar;
                }
                const int32_t hs = d == 0 ? s0 : (d == 1 ? s1 : s2), he = d == 0 ? e0 : (d == 1 ? e1 : e2), hr = d == 0 ? r0 : (d == 1 ? r1 : r2);
                R[k] = make_int4(pm, ar, hs, he);
                Q[k] = make_int4(hr, l1v, l1r, l2v);
            }
        }
    }
    // The lanes this line cannot settle -- hi outside the line (a third row of the bin below the probe's end, a probe outside its
    // contig's table) or an overlap whose first row lies below the two prefix-max levels the record carries -- are NOT finished here:
    // a few per cent of the lanes, but some in nearly every wavefront, and each would put one or more dependent fabric round trips into
    // the wavefront's lifetime (measured: 1.9 instead of 1.0 ms for config 4).  They are marked in a bit mask per (wavefront, item) and
    // k_nearest_k1_rest finishes them with the two-gather form.
    // A third class (round 5): an overlap below the record's two levels is SETTLED except for its build row -- distance 0, found 1 --
    // and what the bound search needs is in registers here.  Those probes leave {hi-bound, segment start << 32 | probe start} in their
    // own result slots (out_idx / out_dist) and a bit in the second mask: the rest kernel picks the search up from two reads instead
    // of redoing the probe from its three columns and the start table (1.8 % of config 4's probes, 70 % of the leftovers).
    int32_t idx[N], found[N];
    long long dist[N];
    bool slow[N], deep[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        idx[k] = -1; dist[k] = -1; found[k] = 0;
        slow[k] = far[k]; deep[k] = false;
        if (i0 + k < n && b[k] > a[k] && !far[k]) {
            const bool have_l = hi[k] > a[k], have_r = hi[k] < b[k];
            if (have_l && lt_op<STRICT>(s[k], R[k].x)) {
                dist[k] = 0; found[k] = 1;
                if (!(Q[k].z >= 0 && lt_op<STRICT>(s[k], Q[k].y))) idx[k] = R[k].y;
                else if (!lt_op<STRICT>(s[k], Q[k].w)) idx[k] = Q[k].z;
                else {
                    deep[k] = true;
                    idx[k] = hi[k];
                    dist[k] = (long long)(((unsigned long long)(unsigned int)a[k] << 32) | (unsigned long long)(unsigned int)s[k]);
                }
            } else {
                const long long dl = (long long)s[k] - (long long)R[k].x;
                const long long dr = have_r ? gap_dist(s[k], e[k], R[k].z, R[k].w) : 0;
                if (have_l && (!have_r || dl <= dr)) { idx[k] = R[k].y; dist[k] = dl; found[k] = 1; }
                else if (have_r) { idx[k] = Q[k].x; dist[k] = dr; found[k] = 1; }
            }
        }
    }
    // rest[N * wavefront + k], bit l <=> item k of lane l is left over (a single list cursor would serialise 390 k same-address atomics)
    {
        const int lane = threadIdx.x & (kWave - 1);
        const int64_t wf = ((int64_t)blockIdx.x * PROBE_THREADS + threadIdx.x) / kWave;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const unsigned long long m = __ballot(slow[k]), m2 = __ballot(deep[k]);
            if (lane == 0) { rest[N * wf + k] = m; rest[n_words + N * wf + k] = m2; }
        }
    }
    if (N == 2 && i0 + 2 <= n && ((reinterpret_cast<uintptr_t>(out_idx) | reinterpret_cast<uintptr_t>(out_n)) & 7u) == 0 &&
        (reinterpret_cast<uintptr_t>(out_dist) & 15u) == 0) {
        typedef long long v2ll __attribute__((ext_vector_type(2)));
        typedef int v2i __attribute__((ext_vector_type(2)));
        v2i vi; vi.x = idx[0]; vi.y = idx[N - 1];
        v2i vf; vf.x = found[0]; vf.y = found[N - 1];
        v2ll vd; vd.x = dist[0]; vd.y = dist[N - 1];
        __builtin_nontemporal_store(vi, reinterpret_cast<v2i*>(out_idx + i0));
        __builtin_nontemporal_store(vd, reinterpret_cast<v2ll*>(out_dist + i0));
        __builtin_nontemporal_store(vf, reinterpret_cast<v2i*>(out_n + i0));
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (i0 + k < n) { out_idx[i0 + k] = idx[k]; out_dist[i0 + k] = dist[k]; out_n[i0 + k] = found[k]; }
    }
}

// The probes k_nearest_k1_lines left over, finished with the two-gather form: start table, nrec, and the bound search where the record's
// two levels do not reach.  Nearly every wavefront of the lines kernel leaves a few, so they are COMPACTED first: a workgroup reads
// REST_WORDS mask words (REST_WORDS x 64 probes), lists the marked probes in LDS (exclusive scan of the popcounts) and works through the
// list with full wavefronts -- left in place, ~ 3 % of the lanes kept every wavefront alive for the longest dependent chain (0.81 ms).
constexpr int REST_WORDS = 128;
template <bool STRICT, int N>
__global__ __launch_bounds__(PROBE_THREADS) void k_nearest_k1_rest(IndexView ix, const int32_t* __restrict__ pc, const int32_t* __restrict__ ps,
                                                                   const int32_t* __restrict__ pe, int64_t n, int64_t n_words,
                                                                   const unsigned long long* __restrict__ rest, int32_t* __restrict__ out_idx,
                                                                   long long* __restrict__ out_dist, int32_t* __restrict__ out_n) {
    __shared__ unsigned short l_list[REST_WORDS * 64];                        // (word of the workgroup) << 6 | lane: 16 KB, eight workgroups per CU
    __shared__ int l_wsum[PROBE_THREADS / kWave];
    static_assert(REST_WORDS * 64 <= 0x2000 && (REST_WORDS & (REST_WORDS - 1)) == 0, "a list entry: class bit 13, word 7 bits, lane 6 bits");
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
    const int64_t w = (int64_t)blockIdx.x * REST_WORDS + tid;
    unsigned long long m = (tid < REST_WORDS && w < n_words) ? rest[w] : 0ull;
    unsigned long long m2 = (tid < REST_WORDS && w < n_words) ? rest[n_words + w] : 0ull;      // settled but for the build row (see the lines kernel)
    const int cnt = __popcll(m) + __popcll(m2);
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) { const int t = __shfl_up(inc, d, kWave); if (lane >= d) inc += t; }
    if (lane == kWave - 1) l_wsum[wv] = inc;
    __syncthreads();
    int pre = inc - cnt, total = 0;
#pragma unroll
    for (int i = 0; i < PROBE_THREADS / kWave; ++i) { const int x = l_wsum[i]; if (i < wv) pre += x; total += x; }
    // word w = N * wavefront + item; bit l = lane l of that wavefront; the lines kernel's probe of (wavefront, lane, item) = (64 wavefront + l) N + item
    while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        l_list[pre++] = (unsigned short)((tid << 6) | l);
    }
    while (m2) {
        const int l = __builtin_ctzll(m2);
        m2 &= m2 - 1;
        l_list[pre++] = (unsigned short)(0x2000 | (tid << 6) | l);
    }
    __syncthreads();
    for (int q = tid; q < total; q += PROBE_THREADS) {
        const int ent = l_list[q];
        const int64_t wq = (int64_t)blockIdx.x * REST_WORDS + ((ent >> 6) & (REST_WORDS - 1));
        const int64_t i = (wq / N) * (int64_t)(kWave * N) + (wq % N) + (int64_t)(ent & 63) * N;
        if (i >= n) continue;                                                  // (defensive: the lines kernel never marks a row beyond n)
        if (ent & 0x2000) {
            const int hi2 = out_idx[i];
            const unsigned long long st = (unsigned long long)out_dist[i];
            out_idx[i] = ix.b_row[bound_lo_near<STRICT>(ix, (int)(st >> 32), hi2, (int32_t)(uint32_t)st)];
            out_dist[i] = 0;
            continue;
        }
        int32_t c[1] = {pc[i]}, s = ps[i], e[1] = {pe[i]};
        bool valid[1] = {true};
        int a[1], b[1], hi[1];
        bound_hi_tab4<STRICT>(ix, c, valid, e, a, b, hi);
        int32_t idx = -1; long long dist = -1; int32_t found = 0;
        if (b[0] > a[0]) {
            const int4 R = ix.nrec[2 * (int64_t)hi[0]], Q = ix.nrec[2 * (int64_t)hi[0] + 1];
            nearest_k1_resolve<STRICT>(ix, a[0], b[0], hi[0], s, e[0], R, Q, idx, dist, found);
        }
        out_idx[i] = idx; out_dist[i] = dist; out_n[i] = found;
    }
}

// General k / include_overlaps: per-probe merge of three ordered streams (overlapping rows in
// (start,row) order; "left" rows by end descending; "right" rows by start ascending).
// One thread per probe; k slots per probe, unused slots -1.
template <bool STRICT>
__global__ __launch_bounds__(PROBE_THREADS) void k_nearest_general(IndexView ix, const int32_t* __restrict__ pc,
                                                                   const int32_t* __restrict__ ps,
                                                                   const int32_t* __restrict__ pe, int64_t n, int kk,
                                                                   int include_overlaps, const int32_t* __restrict__ out_row,
                                                                   int32_t* __restrict__ out_idx,
                                                                   long long* __restrict__ out_dist,
                                                                   int32_t* __restrict__ out_n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t qs = ps[i], qe = pe[i];
    const int64_t o = out_row ? (int64_t)out_row[i] : i;
    int32_t* oi = out_idx + o * kk;
    long long* od = out_dist + o * kk;
    for (int r = 0; r < kk; ++r) { oi[r] = -1; od[r] = -1; }
    int a, b;
    seg_bounds(ix, pc[i], true, a, b);
    int found = 0;
    if (b > a) {
        const int hi = bound_hi<STRICT>(ix, a, b, qe);
        if (include_overlaps) {
            // the first kk overlapping rows in (start, row) order, over the block maxima of the ends: a contig-wide row at the
            // contig's start does not make this a scan of every row below the probe
            const int lo = bound_lo<STRICT>(ix, a, hi, qs);
            hier_walk_up<STRICT>(ix.hier, lo, hi, qs, [&](int p) { oi[found] = ix.b_row[p]; od[found] = 0; ++found; return found < kk; });
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
    out_n[o] = found;
}

}  // namespace ivj
