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
template <bool STRICT>
__global__ __launch_bounds__(PROBE_THREADS) void k_coverage(const int32_t* __restrict__ seg, const uint32_t* __restrict__ cid1,
                                                            const int32_t* __restrict__ m_start, const int32_t* __restrict__ m_end,
                                                            const long long* __restrict__ pl, int32_t n_contigs,
                                                            const int32_t* __restrict__ pc, const int32_t* __restrict__ ps,
                                                            const int32_t* __restrict__ pe, int64_t n, long long* __restrict__ cov) {
    const int64_t i = (int64_t)blockIdx.x * PROBE_THREADS + threadIdx.x;
    if (i >= n) return;
    const int32_t c = pc[i];
    long long out = 0;
    if ((uint32_t)c < (uint32_t)n_contigs) {
        const int a = seg[c], b = seg[c + 1];
        const long long qs = ps[i], qe = (long long)pe[i] + (STRICT ? 0 : 1);
        if (b > a && qe > qs) {
            const int j0 = (int)cid1[a] - 1, j1 = (int)cid1[b - 1];
            int lo = j0, hi = j1;                              // first: first cluster whose half-open end > qs
            while (lo < hi) { const int m = lo + ((hi - lo) >> 1); if ((long long)m_end[m] + (STRICT ? 0 : 1) > qs) hi = m; else lo = m + 1; }
            const int first = lo;
            lo = first; hi = j1;                               // last: first cluster whose start >= qe
            while (lo < hi) { const int m = lo + ((hi - lo) >> 1); if ((long long)m_start[m] >= qe) hi = m; else lo = m + 1; }
            const int last = lo;
            if (last > first) {
                auto clipped = [&](int j) -> long long {
                    const long long s = m_start[j], e = (long long)m_end[j] + (STRICT ? 0 : 1);
                    const long long l = (e < qe ? e : qe) - (s > qs ? s : qs);
                    return l > 0 ? l : 0;
                };
                if (last - first == 1) out = clipped(first);
                else out = clipped(first) + clipped(last - 1) + (pl[last - 1] - pl[first + 1]);
            }
        }
    }
    cov[i] = out;
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

// first / last union interval touching [ls, le') of a left row, and the number of pieces that remain
template <bool STRICT>
__device__ __forceinline__ int subtract_span(const int32_t* __restrict__ seg, const uint32_t* __restrict__ cid1,
                                             const uint32_t* __restrict__ newidx, const long long* __restrict__ u_start,
                                             const long long* __restrict__ u_end, int32_t n_contigs, int32_t c, long long ls,
                                             long long le, int& first, int& last) {
    first = 0; last = 0;
    if (le <= ls) return 0;                                   // the row holds no position
    if ((uint32_t)c >= (uint32_t)n_contigs) return 1;
    const int a = seg[c], b = seg[c + 1];
    if (b <= a) return 1;
    const int j0 = (int)newidx[cid1[a] - 1u], j1 = (int)newidx[cid1[b - 1]];
    int lo = j0, hi = j1;
    while (lo < hi) { const int m = lo + ((hi - lo) >> 1); if (u_end[m] > ls) hi = m; else lo = m + 1; }
    first = lo;
    hi = j1;
    while (lo < hi) { const int m = lo + ((hi - lo) >> 1); if (u_start[m] >= le) hi = m; else lo = m + 1; }
    last = lo;
    if (last == first) return 1;
    return (u_start[first] > ls ? 1 : 0) + (last - first - 1) + (u_end[last - 1] < le ? 1 : 0);
}

template <bool STRICT>
__global__ __launch_bounds__(PROBE_THREADS) void k_subtract_count(const int32_t* __restrict__ seg, const uint32_t* __restrict__ cid1,
                                                                  const uint32_t* __restrict__ newidx,
                                                                  const long long* __restrict__ u_start, const long long* __restrict__ u_end,
                                                                  int32_t n_contigs, const int32_t* __restrict__ lc,
                                                                  const int32_t* __restrict__ lstart, const int32_t* __restrict__ lend,
                                                                  int64_t n, long long* __restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * PROBE_THREADS + threadIdx.x;
    if (i >= n) return;
    int first, last;
    cnt[i] = subtract_span<STRICT>(seg, cid1, newidx, u_start, u_end, n_contigs, lc[i], (long long)lstart[i],
                                   (long long)lend[i] + (STRICT ? 0 : 1), first, last);
}

// off = exclusive sum scan of cnt; pieces of one left row are written in ascending order
template <bool STRICT>
__global__ __launch_bounds__(PROBE_THREADS) void k_subtract_fill(const int32_t* __restrict__ seg, const uint32_t* __restrict__ cid1,
                                                                 const uint32_t* __restrict__ newidx,
                                                                 const long long* __restrict__ u_start, const long long* __restrict__ u_end,
                                                                 int32_t n_contigs, const int32_t* __restrict__ lc,
                                                                 const int32_t* __restrict__ lstart, const int32_t* __restrict__ lend,
                                                                 const int32_t* __restrict__ row_id, int64_t n,
                                                                 const long long* __restrict__ off, int32_t* __restrict__ o_row,
                                                                 int32_t* __restrict__ o_start, int32_t* __restrict__ o_end) {
    const int64_t i = (int64_t)blockIdx.x * PROBE_THREADS + threadIdx.x;
    if (i >= n) return;
    const long long ls = lstart[i], le = (long long)lend[i] + (STRICT ? 0 : 1);
    int first, last;
    const int k = subtract_span<STRICT>(seg, cid1, newidx, u_start, u_end, n_contigs, lc[i], ls, le, first, last);
    if (k == 0) return;
    long long o = off[i];
    const int32_t r = row_id ? row_id[i] : (int32_t)i;
    auto emit = [&](long long s, long long e) {
        o_row[o] = r; o_start[o] = (int32_t)s; o_end[o] = (int32_t)(e - (STRICT ? 0 : 1)); ++o;
    };
    if (last == first) { emit(ls, le); return; }
    if (u_start[first] > ls) emit(ls, u_start[first]);
    for (int j = first; j + 1 < last; ++j) emit(u_end[j], u_start[j + 1]);
    if (u_end[last - 1] < le) emit(u_end[last - 1], le);
}

}  // namespace ivj
