// sortscan.hip.h -- the sort-scan family on top of the sorted build index:
// pb.merge / pb.cluster (one frame) and pb.coverage (bases of every probe interval covered by the union of
// the build side).  Reference: MergeProvider / ClusterProvider / CountOverlapsProvider(coverage = true),
// call sites src/operation.rs:352-418, 306-350; the behaviour the reference's tests pin is listed in DESIGN.md.  Everything here is a pass over the index arrays of index_view.hip.h -- the (contig, start)
// order and the prefix max of the ends are exactly what a sweep needs:
//   a row starts a new cluster  <=>  first row of its contig, or !(start (<) prefix max of the rows before + min_dist)
// with (<) = "<" for Strict (0-based half-open) and "<=" for Weak (1-based closed) coordinates.
#pragma once
#include "index_view.hip.h"

namespace ivj {

template <bool STRICT>
__global__ void k_cluster_flags(const int32_t* __restrict__ b_start, const int2* __restrict__ ep,
                                const int32_t* __restrict__ b_contig, int64_t n, long long min_dist,
                                uint32_t* __restrict__ flags) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    bool first = p == 0 || b_contig[p] != b_contig[p - 1];
    if (!first) {
        const long long lim = (long long)ep[p - 1].y + min_dist;
        const long long s = (long long)b_start[p];
        first = !(STRICT ? (s < lim) : (s <= lim));
    }
    flags[p] = first ? 1u : 0u;
}

// cid1 = inclusive sum scan of flags (1-based cluster id per sorted position).  Per cluster: contig, start
// (start of its first row), end (prefix max at its last row), position of its first row.
__global__ void k_cluster_bounds(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ cid1,
                                 const int32_t* __restrict__ b_start, const int2* __restrict__ ep,
                                 const int32_t* __restrict__ b_contig, int64_t n, int32_t n_contigs,
                                 int32_t* __restrict__ m_contig, int32_t* __restrict__ m_start,
                                 int32_t* __restrict__ m_end, int32_t* __restrict__ m_first) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t c = cid1[p] - 1u;
    if (flags[p]) {
        const int32_t ct = b_contig[p];
        m_contig[c] = ct < n_contigs ? ct : -1;            // rows outside the dictionary cluster among themselves
        m_start[c] = b_start[p];
        m_first[c] = (int32_t)p;
    }
    if (p == n - 1 || flags[p + 1]) m_end[c] = ep[p].y;
    if (p == n - 1) m_first[c + 1] = (int32_t)n;
}

__global__ void k_cluster_counts(const int32_t* __restrict__ m_first, int64_t n_clusters, long long* __restrict__ m_count) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_clusters) m_count[c] = (long long)(m_first[c + 1] - m_first[c]);
}

// pb.cluster: per INPUT row (b_row = original row of sorted position p) the cluster id and bounds
__global__ void k_cluster_scatter(const int32_t* __restrict__ b_row, const uint32_t* __restrict__ cid1,
                                  const int32_t* __restrict__ m_start, const int32_t* __restrict__ m_end, int64_t n,
                                  long long* __restrict__ out_cluster, int32_t* __restrict__ out_start,
                                  int32_t* __restrict__ out_end) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t c = cid1[p] - 1u;
    const int32_t r = b_row[p];
    out_cluster[r] = (long long)c;
    out_start[r] = m_start[c];
    out_end[r] = m_end[c];
}

// half-open length of every merged interval (Weak: closed [s, e] = [s, e + 1)), clamped at 0 for clusters
// made of inverted rows only; an exclusive sum scan of it gives the covered bases before each cluster
template <bool STRICT>
__global__ void k_merged_lengths(const int32_t* __restrict__ m_start, const int32_t* __restrict__ m_end, int64_t n_clusters,
                                 long long* __restrict__ len) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_clusters) return;
    const long long l = (long long)m_end[c] + (STRICT ? 0 : 1) - (long long)m_start[c];
    len[c] = l > 0 ? l : 0;
}

// pb.coverage: the clusters of the probe's contig are disjoint and sorted, so the covered bases of [qs, qe') are
// the prefix-sum difference over the clusters that intersect it, minus what the two end clusters stick out.
//   first = first cluster with end' > qs,   last = first cluster with start >= qe'      (end' = half-open end)
// Both come from the direct-address table over the starts (index_view.hip.h) instead of a bound search over
// the cluster arrays: with p1 = #rows starting <= qs, the cluster of row p1 - 1 is the only one that can reach
// past qs (earlier ones end before it starts); with p2 = #rows starting < qe', last = cluster of row p2 - 1, + 1.
constexpr int COV_ITEMS = 2;

template <bool STRICT>
__global__ __launch_bounds__(PROBE_THREADS) void k_coverage(IndexView ix, const uint32_t* __restrict__ cid1,
                                                            const int32_t* __restrict__ m_start, const int32_t* __restrict__ m_end,
                                                            const long long* __restrict__ pl,
                                                            const int32_t* __restrict__ pc, const int32_t* __restrict__ ps,
                                                            const int32_t* __restrict__ pe, const int32_t* __restrict__ out_row,
                                                            int64_t n, bool vec_ok, long long* __restrict__ cov) {
    // out_row: the probes are a bucketed permutation (partition.hip.h); the result goes to the original row
    const long long ntiles = (n + PROBE_THREADS * COV_ITEMS - 1) / (PROBE_THREADS * COV_ITEMS);
    const long long tile = xcd_tile64(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int64_t i0 = ((int64_t)tile * PROBE_THREADS + threadIdx.x) * COV_ITEMS;
    if (i0 >= n) return;
    int32_t c[COV_ITEMS], s[COV_ITEMS], e[COV_ITEMS];
    load_items(pc, i0, n, vec_ok, -1, c);
    load_items(ps, i0, n, vec_ok, 0, s);
    load_items(pe, i0, n, vec_ok, 0, e);
    bool valid[COV_ITEMS];
    unsigned long long t1[COV_ITEMS], t2[COV_ITEMS];
#pragma unroll
    for (int k = 0; k < COV_ITEMS; ++k) {
        valid[k] = i0 + k < n;
        t1[k] = (unsigned long long)flip(s[k]) + 1ull;                       // first row with start > qs
        t2[k] = (unsigned long long)flip(e[k]) + (STRICT ? 0ull : 1ull);     // first row with start >= qe'
    }
    int a[COV_ITEMS], b[COV_ITEMS], p1[COV_ITEMS], p2[COV_ITEMS];
    lb_tab4(ix.cmeta, ix.brec, ix.bins, ix.use_rec != 0, ix.b_start, ix.n_contigs, c, valid, t1, a, b, p1);
    lb_tab4(ix.cmeta, ix.brec, ix.bins, ix.use_rec != 0, ix.b_start, ix.n_contigs, c, valid, t2, a, b, p2);
#pragma unroll
    for (int k = 0; k < COV_ITEMS; ++k) {
        if (!valid[k]) continue;
        long long out = 0;
        const long long qs = s[k], qe = (long long)e[k] + (STRICT ? 0 : 1);
        if (b[k] > a[k] && qe > qs) {
            const int j0 = (int)cid1[a[k]] - 1;
            int first = j0, last = j0;
            if (p1[k] > a[k]) {
                const int cl = (int)cid1[p1[k] - 1] - 1;
                first = ((long long)m_end[cl] + (STRICT ? 0 : 1) > qs) ? cl : cl + 1;
            }
            if (p2[k] > a[k]) last = (int)cid1[p2[k] - 1];
            if (last > first) {
                auto clipped = [&](int j) -> long long {
                    const long long ms = m_start[j], me = (long long)m_end[j] + (STRICT ? 0 : 1);
                    const long long l = (me < qe ? me : qe) - (ms > qs ? ms : qs);
                    return l > 0 ? l : 0;
                };
                if (last - first == 1) out = clipped(first);
                else out = clipped(first) + clipped(last - 1) + (pl[last - 1] - pl[first + 1]);
            }
        }
        cov[out_row ? (int64_t)out_row[i0 + k] : i0 + k] = out;
    }
}

// ---- subtract / complement -------------------------------------------------------------------------
// Both are "an interval minus the union of the other side": subtract(df1, df2) per df1 row,
// complement(df, view) = subtract(view, df).  The union is the cluster sweep with min_dist = 1 (bookended
// half-open intervals, adjacent closed ones, leave no position between them), compacted to the clusters
// that hold at least one position; everything is carried as half-open int64 [s, e').

template <bool STRICT>
__global__ void k_union_flags(const int32_t* __restrict__ m_start, const int32_t* __restrict__ m_end, int64_t n_clusters,
                              uint32_t* __restrict__ keep) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_clusters) keep[c] = ((long long)m_end[c] + (STRICT ? 0 : 1) > (long long)m_start[c]) ? 1u : 0u;
}

// newidx = exclusive sum scan of keep (n_clusters + 1 entries: the last one is the number of kept clusters)
template <bool STRICT>
__global__ void k_union_compact(const int32_t* __restrict__ m_start, const int32_t* __restrict__ m_end,
                                const uint32_t* __restrict__ keep, const uint32_t* __restrict__ newidx, int64_t n_clusters,
                                long long* __restrict__ u_start, long long* __restrict__ u_end) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_clusters && keep[c]) {
        u_start[newidx[c]] = (long long)m_start[c];
        u_end[newidx[c]] = (long long)m_end[c] + (STRICT ? 0 : 1);
    }
}

// first / last union interval touching [ls, le') of a left row, and the number of pieces that remain.
// Same table lookups as k_coverage; keep / newidx translate a cluster to its place among the kept ones
// (newidx of a dropped cluster = index of the next kept one).
template <bool STRICT>
__device__ __forceinline__ int subtract_span(const IndexView& ix, const uint32_t* __restrict__ cid1, const uint32_t* __restrict__ keep,
                                             const uint32_t* __restrict__ newidx, const long long* __restrict__ u_start,
                                             const long long* __restrict__ u_end, int32_t c, int32_t s, int32_t e, int& first, int& last) {
    first = 0; last = 0;
    const long long ls = s, le = (long long)e + (STRICT ? 0 : 1);
    if (le <= ls) return 0;                                   // the row holds no position
    const int32_t cc[1] = {c};
    const bool valid[1] = {true};
    const unsigned long long t1[1] = {(unsigned long long)flip(s) + 1ull};
    const unsigned long long t2[1] = {(unsigned long long)flip(e) + (STRICT ? 0ull : 1ull)};
    int a[1], b[1], p1[1], p2[1];
    lb_tab4(ix.cmeta, ix.brec, ix.bins, ix.use_rec != 0, ix.b_start, ix.n_contigs, cc, valid, t1, a, b, p1);
    lb_tab4(ix.cmeta, ix.brec, ix.bins, ix.use_rec != 0, ix.b_start, ix.n_contigs, cc, valid, t2, a, b, p2);
    if (b[0] <= a[0]) return 1;
    const int j0 = (int)newidx[cid1[a[0]] - 1u];
    first = j0; last = j0;
    if (p1[0] > a[0]) {
        const uint32_t cl = cid1[p1[0] - 1] - 1u;
        const int u = (int)newidx[cl];
        first = (keep[cl] && u_end[u] > ls) ? u : (int)newidx[cl + 1u];
    }
    if (p2[0] > a[0]) last = (int)newidx[cid1[p2[0] - 1]];
    if (last <= first) { last = first; return 1; }
    return (u_start[first] > ls ? 1 : 0) + (last - first - 1) + (u_end[last - 1] < le ? 1 : 0);
}

template <bool STRICT>
__global__ __launch_bounds__(PROBE_THREADS) void k_subtract_count(IndexView ix, const uint32_t* __restrict__ cid1,
                                                                  const uint32_t* __restrict__ keep, const uint32_t* __restrict__ newidx,
                                                                  const long long* __restrict__ u_start, const long long* __restrict__ u_end,
                                                                  const int32_t* __restrict__ lc,
                                                                  const int32_t* __restrict__ lstart, const int32_t* __restrict__ lend,
                                                                  const int32_t* __restrict__ pos_row, int64_t n,
                                                                  long long* __restrict__ cnt) {
    // pos_row: the rows are a bucketed permutation; counts (and offsets) are kept in ORIGINAL row order so
    // that the pieces come out ordered by left row whatever order the kernel visits the rows in
    const int64_t i = (int64_t)blockIdx.x * PROBE_THREADS + threadIdx.x;
    if (i >= n) return;
    int first, last;
    cnt[pos_row ? (int64_t)pos_row[i] : i] = subtract_span<STRICT>(ix, cid1, keep, newidx, u_start, u_end, lc[i], lstart[i], lend[i], first, last);
}

// off = exclusive sum scan of cnt; pieces of one left row are written in ascending order
template <bool STRICT>
__global__ __launch_bounds__(PROBE_THREADS) void k_subtract_fill(IndexView ix, const uint32_t* __restrict__ cid1,
                                                                 const uint32_t* __restrict__ keep, const uint32_t* __restrict__ newidx,
                                                                 const long long* __restrict__ u_start, const long long* __restrict__ u_end,
                                                                 const int32_t* __restrict__ lc,
                                                                 const int32_t* __restrict__ lstart, const int32_t* __restrict__ lend,
                                                                 const int32_t* __restrict__ pos_row, const int32_t* __restrict__ row_id,
                                                                 int64_t n, const long long* __restrict__ off, int32_t* __restrict__ o_row,
                                                                 int32_t* __restrict__ o_start, int32_t* __restrict__ o_end) {
    const int64_t i = (int64_t)blockIdx.x * PROBE_THREADS + threadIdx.x;
    if (i >= n) return;
    const long long ls = lstart[i], le = (long long)lend[i] + (STRICT ? 0 : 1);
    int first, last;
    const int k = subtract_span<STRICT>(ix, cid1, keep, newidx, u_start, u_end, lc[i], lstart[i], lend[i], first, last);
    if (k == 0) return;
    const int64_t orig = pos_row ? (int64_t)pos_row[i] : i;
    long long o = off[orig];
    const int32_t r = row_id ? row_id[orig] : (int32_t)orig;
    auto emit = [&](long long s, long long e) {
        o_row[o] = r; o_start[o] = (int32_t)s; o_end[o] = (int32_t)(e - (STRICT ? 0 : 1)); ++o;
    };
    if (last == first) { emit(ls, le); return; }
    if (u_start[first] > ls) emit(ls, u_start[first]);
    for (int j = first; j + 1 < last; ++j) emit(u_end[j], u_start[j + 1]);
    if (u_end[last - 1] < le) emit(u_end[last - 1], le);
}

}  // namespace ivj
