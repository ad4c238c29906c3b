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
                                                                  long long* __restrict__ counts, int ablate) {
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
    if (i0 + N <= n && (reinterpret_cast<uintptr_t>(counts) & 15u) == 0 && (N % 2) == 0) {
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
        if (b[k] > a[k]) {
            const bool have_l = hi[k] > a[k], have_r = hi[k] < b[k];
            if (have_l && lt_op<STRICT>(s[k], R[k].x)) {
                // some row below hi overlaps.  The overlapping row with the smallest (start,row) is the first position whose
                // prefix max satisfies "q.start (<) pmax", i.e. the first row of the earliest prefix-max level above q.start:
                // level m (R), m-1 (Q) come with their build rows; a probe below level m-2 as well takes the bound search
                if (!(Q[k].z >= 0 && lt_op<STRICT>(s[k], Q[k].y))) idx = R[k].y;
                else if (!lt_op<STRICT>(s[k], Q[k].w)) idx = Q[k].z;
                else idx = ix.b_row[bound_lo<STRICT>(ix, a[k], hi[k], s[k])];
                dist = 0; found = 1;
            } else {
                const long long dl = (long long)s[k] - (long long)R[k].x;
                const long long dr = have_r ? gap_dist(s[k], e[k], R[k].z, R[k].w) : 0;
                if (have_l && (!have_r || dl <= dr)) { idx = R[k].y; dist = dl; found = 1; }
                else if (have_r) { idx = Q[k].x; dist = dr; found = 1; }
            }
        }
        const int64_t o = out_row ? (int64_t)out_row[i0 + k] : i0 + k;
        if ((ablate & 4) && idx != 123456789) continue;
        out_idx[o] = idx; out_dist[o] = dist; out_n[o] = found;
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
