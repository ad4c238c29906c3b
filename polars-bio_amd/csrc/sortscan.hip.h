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

// ---- coverage from a grid over the UNION intervals (round 2) ------------------------------------------------------------
// k_coverage above needs ~10 dependent gathers per probe (two table lookups, cluster ids, cluster bounds, prefix sums) and
// therefore bucketed probes + an inverse permutation.  The per-probe kernels are priced per L2 request (~6.5 ps per gather
// level and probe), so the coverage is restated as ONE function C(x) = covered positions below x, answered
// by ONE 16-byte record per endpoint:  coverage([s, e')) = C(e') - C(s).
//   The clusters of a contig (min_dist 0) are disjoint and sorted: S_k = start, E_k = max(start, end') (a cluster that holds no
//   position covers nothing), E_k <= S_(k+1).  Coordinates are handled as ux = x + 2^31 (0 .. 2^32).  Per contig a uniform grid
//   over [S_first, E_last] with <= 2 bins per cluster; the record of the bin at x0:
//     {C(x0) mod 2^32, t1 | t2 << 16, t3 | flags << 16, k0}     k0 = first cluster with E > x0
//   t1..t3 = the first three positions inside the bin where "covered" toggles, as 16-bit offsets from x0 (0xffff: none);
//   flags: 1 = x0 is covered, 2 = the bin has a fourth toggle, 4 = bins wider than 2^16 (no offsets: always search).
//   C(x) for x in the bin = C(x0) + the covered part of [x0, x) from the toggles; only a bin with more than three toggles
//   below x (or a wide grid) searches the cluster arrays from k0.
struct CovMeta { const int4* cm; const int4* rec; };        // cm[2c] = {ulo, uhi, shift, tb}, cm[2c+1] = {C(lo), C(hi), ca, cb}

__device__ __forceinline__ unsigned long long cov_ux(long long x) { return (unsigned long long)(x + 2147483648ll); }

template <bool STRICT>
__device__ __forceinline__ unsigned long long cov_E(const int32_t* __restrict__ m_start, const int32_t* __restrict__ m_end, int k) {
    const long long s = m_start[k], e = (long long)m_end[k] + (STRICT ? 0 : 1);
    return cov_ux(e > s ? e : s);
}

// exact C(x) by search over the clusters [k_lo, cb) of the contig (64-bit): fallback of the grid and the builder's definition
template <bool STRICT>
__device__ __forceinline__ unsigned long long cov_search(const int32_t* __restrict__ m_start, const int32_t* __restrict__ m_end,
                                                         const long long* __restrict__ pl, int k_lo, int cb, unsigned long long x, int* k_out) {
    int lo = k_lo, step = 1;                                                   // first k in [k_lo, cb) with E_k > x: gallop, then bound search
    while (lo + step - 1 < cb && cov_E<STRICT>(m_start, m_end, lo + step - 1) <= x) { lo += step; step <<= 1; }
    int hi = lo + step - 1 < cb ? lo + step - 1 : cb;
    while (lo < hi) { const int m = lo + ((hi - lo) >> 1); if (cov_E<STRICT>(m_start, m_end, m) <= x) lo = m + 1; else hi = m; }
    if (k_out) *k_out = lo;
    if (lo >= cb) return (unsigned long long)pl[cb];
    const unsigned long long S = cov_ux(m_start[lo]);
    return (unsigned long long)pl[lo] + (x > S ? x - S : 0ull);
}

// one thread per contig: cluster range, grid geometry, C at both ends
template <bool STRICT>
__global__ void k_cov_meta(const int32_t* __restrict__ seg, const uint32_t* __restrict__ cid1, const int32_t* __restrict__ m_start,
                           const int32_t* __restrict__ m_end, const long long* __restrict__ pl, int32_t n_contigs, int4* __restrict__ cm) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_contigs) return;
    const int a = seg[c], b = seg[c + 1];
    // clusters of the contigs before this one: also the slot base of a contig WITHOUT rows, so that the slot bases stay
    // monotonic (k_cov_records finds a slot's contig by a bound search over them) whatever ids the probe-only chroms have
    int ca = a > 0 ? (int)cid1[a - 1] : 0, cb = ca, shift = 0;
    uint32_t ulo = 0, uhi = 0;
    if (b > a) {
        cb = (int)cid1[b - 1];
        const unsigned long long lo = cov_ux(m_start[ca]), hi = cov_E<STRICT>(m_start, m_end, cb - 1);
        ulo = (uint32_t)lo; uhi = hi > 0xffffffffull ? 0xffffffffu : (uint32_t)hi;
        unsigned long long cap = 2ull * (unsigned long long)(cb - ca);
        if (cap < 2) cap = 2;
        while ((((unsigned long long)uhi - ulo) >> shift) + 1ull > cap) ++shift;
    }
    cm[2 * c] = make_int4((int)ulo, (int)uhi, shift, 2 * ca + 2 * c);
    cm[2 * c + 1] = make_int4(b > a ? (int)(uint32_t)pl[ca] : 0, b > a ? (int)(uint32_t)pl[cb] : 0, ca, cb);
}

// one thread per grid slot
template <bool STRICT>
__global__ void k_cov_records(const int4* __restrict__ cm, int32_t n_contigs, int64_t n_slots, const int32_t* __restrict__ m_start,
                              const int32_t* __restrict__ m_end, const long long* __restrict__ pl, int4* __restrict__ rec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    int lo = 0, hi = n_contigs;
    while (lo < hi) { const int m = (lo + hi) >> 1; if ((int64_t)cm[2 * m].w <= i) lo = m + 1; else hi = m; }
    const int c = lo - 1;
    int4 r = make_int4(0, -1, 0xffff | (4 << 16), 0);
    if (c >= 0) {
        const int4 m0 = cm[2 * c], m1 = cm[2 * c + 1];
        const int ca = m1.z, cb = m1.w, shift = m0.z;
        const unsigned long long x0 = (unsigned long long)(uint32_t)m0.x + ((unsigned long long)(i - (int64_t)m0.w) << shift);
        if (cb > ca && x0 <= (unsigned long long)(uint32_t)m0.y) {
            int k0;
            const unsigned long long c0 = cov_search<STRICT>(m_start, m_end, pl, ca, cb, x0, &k0);
            uint32_t flags = 0, t[3] = {0xffffu, 0xffffu, 0xffffu};
            if (shift > 16) flags = 4;
            else {
                const unsigned long long W = 1ull << shift;
                const bool inside = k0 < cb && cov_ux(m_start[k0]) <= x0;
                if (inside) flags |= 1;
                int nt = 0;
                // toggles after x0: (the end of the covering cluster,) then start / end of the following clusters
                for (int k = k0; k < cb && nt < 4; ++k) {
                    const unsigned long long S = cov_ux(m_start[k]), E = cov_E<STRICT>(m_start, m_end, k);
                    if (!(k == k0 && inside)) {
                        if (S - x0 >= W) break;
                        if (nt < 3) t[nt] = (uint32_t)(S - x0);
                        ++nt;
                    }
                    if (E - x0 >= W) break;
                    if (nt < 3) t[nt] = (uint32_t)(E - x0);
                    ++nt;
                }
                if (nt > 3) flags |= 2;
            }
            r = make_int4((int)(uint32_t)c0, (int)(t[0] | (t[1] << 16)), (int)(t[2] | (flags << 16)), k0);
        }
    }
    rec[i] = r;
}

constexpr int COV2_ITEMS = 2;
constexpr int COV2_TILES_PER_WG = 4;

template <bool STRICT, bool LM>
__global__ __launch_bounds__(PROBE_THREADS) void k_coverage_grid(CovMeta g, int32_t n_contigs, const int32_t* __restrict__ m_start,
                                                                 const int32_t* __restrict__ m_end, const long long* __restrict__ pl,
                                                                 const int32_t* __restrict__ pc, const int32_t* __restrict__ ps,
                                                                 const int32_t* __restrict__ pe, int64_t n, bool vec_ok,
                                                                 long long* __restrict__ cov) {
    constexpr int N = COV2_ITEMS;
    __shared__ int4 l_cm[LM ? 2 * CM_LDS : 1];
    if (LM) {
        for (int i = threadIdx.x; i < 2 * n_contigs; i += PROBE_THREADS) l_cm[i] = g.cm[i];
        __syncthreads();
    }
#pragma unroll 1
  for (int t = 0; t < COV2_TILES_PER_WG; ++t) {
    const int64_t i0 = ((int64_t)blockIdx.x * COV2_TILES_PER_WG + t) * (PROBE_THREADS * N) + (int64_t)threadIdx.x * N;
    if (i0 - (int64_t)threadIdx.x * N >= n) break;
    int32_t c[N], s[N], e[N];
    load_items_nt(pc, i0, n, vec_ok, -1, c);
    load_items_nt(ps, i0, n, vec_ok, 0, s);
    load_items_nt(pe, i0, n, vec_ok, 0, e);
    // both endpoints of every probe: state 0 = C(lo), 1 = C(hi), 2 = grid record
    unsigned long long x[2 * N];
    int st[2 * N];
    uint32_t slot[2 * N], dd[2 * N];
    int4 m1[N], rec[2 * N];
    bool wide[N], live[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const bool ok = i0 + k < n && (uint32_t)c[k] < (uint32_t)n_contigs;
        int4 m0 = make_int4(0, 0, 0, 0);
        m1[k] = make_int4(0, 0, 0, 0);
        if (ok) {
            if (LM) { m0 = l_cm[2 * c[k]]; m1[k] = l_cm[2 * c[k] + 1]; }
            else { m0 = g.cm[2 * c[k]]; m1[k] = g.cm[2 * c[k] + 1]; }
        }
        x[2 * k] = cov_ux(s[k]);
        x[2 * k + 1] = cov_ux((long long)e[k] + (STRICT ? 0 : 1));
        live[k] = ok && m1[k].w > m1[k].z && x[2 * k + 1] > x[2 * k];
        wide[k] = m0.z > 16;
        const uint32_t ulo = (uint32_t)m0.x, uhi = (uint32_t)m0.y;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned long long xx = x[2 * k + h];
            st[2 * k + h] = (!live[k] || xx <= ulo) ? 0 : (xx > uhi ? 1 : 2);
            const uint32_t off = (uint32_t)xx - ulo;
            slot[2 * k + h] = (uint32_t)m0.w + (off >> m0.z);
            dd[2 * k + h] = off & ((1u << m0.z) - 1u);                          // shift <= 31 (cap >= 2)
        }
        const bool same = st[2 * k] == 2 && st[2 * k + 1] == 2 && slot[2 * k] == slot[2 * k + 1];
        rec[2 * k] = make_int4(0, 0, 0, 0); rec[2 * k + 1] = make_int4(0, 0, 0, 0);
        if (st[2 * k + 1] == 2) rec[2 * k + 1] = g.rec[slot[2 * k + 1]];
        if (st[2 * k] == 2 && !same) rec[2 * k] = g.rec[slot[2 * k]];
        if (same) st[2 * k] = 3;                                                 // copy of the other endpoint's record
    }
    long long out[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (st[2 * k] == 3) { rec[2 * k] = rec[2 * k + 1]; st[2 * k] = 2; }
        uint32_t cv[2];
        bool slow = live[k] && (x[2 * k + 1] - x[2 * k]) > 0xffffffffull;       // the whole int32 range under Weak: 64-bit path
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = 2 * k + h;
            if (st[q] == 0) cv[h] = (uint32_t)m1[k].x;
            else if (st[q] == 1) cv[h] = (uint32_t)m1[k].y;
            else {
                const uint32_t w1 = (uint32_t)rec[q].y, w2 = (uint32_t)rec[q].z, fl = w2 >> 16, d = dd[q];
                const uint32_t t1 = w1 & 0xffffu, t2 = w1 >> 16, t3 = w2 & 0xffffu;
                const uint32_t a0 = t1 < d ? t1 : d, a1 = t2 < d ? t2 : d, a2 = t3 < d ? t3 : d;
                const uint32_t part = (fl & 1u) ? a0 + (a2 - a1) : (a1 - a0) + (d - a2);
                cv[h] = (uint32_t)rec[q].x + part;
                if ((fl & 4u) || ((fl & 2u) && d > t3))
                    cv[h] = (uint32_t)cov_search<STRICT>(m_start, m_end, pl, rec[q].w > m1[k].z ? rec[q].w : m1[k].z, m1[k].w, x[q], nullptr);
            }
        }
        out[k] = live[k] ? (long long)(uint32_t)(cv[1] - cv[0]) : 0ll;
        if (slow) out[k] = (long long)(cov_search<STRICT>(m_start, m_end, pl, m1[k].z, m1[k].w, x[2 * k + 1], nullptr) -
                                       cov_search<STRICT>(m_start, m_end, pl, m1[k].z, m1[k].w, x[2 * k], nullptr));
        (void)wide;
    }
    if (i0 + N <= n && (reinterpret_cast<uintptr_t>(cov) & 15u) == 0 && N == 2) {
        typedef long long v2ll __attribute__((ext_vector_type(2)));
        v2ll v; v.x = out[0]; v.y = out[1];
        __builtin_nontemporal_store(v, reinterpret_cast<v2ll*>(cov + i0));
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) if (i0 + k < n) cov[i0 + k] = out[k];
    }
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

// ---- subtract / complement on a grid over the union intervals (round 2) ---------------------------------------------------
// subtract_span above makes ~10 dependent gathers per left row (two table lookups, cluster ids, kept-cluster indices, interval
// bounds), twice (count pass, fill pass), on bucketed rows.  The same restatement as for coverage: ONE 16-byte record per
// point gives  K(x) = first union interval (u_start / u_end: compacted, disjoint, sorted, non-empty) whose end' is > x,  and
// IN(x) = "x is covered".  For a left row [ls, le') with y = le' - 1:
//     first = K(ls),  last = K(y) + IN(y)      (intervals [first, last) touch the row)
//     pieces = 1 if last <= first, else  !IN(ls) + (last - first - 1) + !IN(y)
// Grid per contig over [S_first, E_last) with <= 2 bins per union interval; record of the bin at x0:
//     {K(x0), t1 | t2 << 16, t3 | flags << 16, -}   t = offsets of the first three toggles inside the bin,
//     flags 1: x0 covered, 2: a fourth toggle exists, 4: bins wider than 2^15 (search from K(x0)).
struct SubGrid { const int4* cm; const int4* rec; };     // cm[2c] = {lo, span lo, span hi, shift}, cm[2c+1] = {tb, ua, ub, 0}

__device__ __forceinline__ void sub_search(const long long* __restrict__ u_start, const long long* __restrict__ u_end, int k_lo, int ub,
                                           long long x, int& K, bool& in) {
    int lo = k_lo, step = 1;
    while (lo + step - 1 < ub && u_end[lo + step - 1] <= x) { lo += step; step <<= 1; }
    int hi = lo + step - 1 < ub ? lo + step - 1 : ub;
    while (lo < hi) { const int m = lo + ((hi - lo) >> 1); if (u_end[m] <= x) lo = m + 1; else hi = m; }
    K = lo;
    in = lo < ub && u_start[lo] <= x;
}

__global__ void k_sub_meta(const int32_t* __restrict__ seg, const uint32_t* __restrict__ cid1, const uint32_t* __restrict__ newidx,
                           const long long* __restrict__ u_start, const long long* __restrict__ u_end, int32_t n_contigs, int4* __restrict__ cm) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_contigs) return;
    const int a = seg[c], b = seg[c + 1];
    // union intervals of the contigs before this one (also for a contig without rows: monotonic slot bases, see k_cov_meta)
    int ua = (int)newidx[a > 0 ? cid1[a - 1] : 0u], ub = ua, shift = 0;
    long long lo = 0;
    unsigned long long span = 0;
    if (b > a) {
        ub = (int)newidx[cid1[b - 1]];
        if (ub > ua) {
            lo = u_start[ua];
            span = (unsigned long long)(u_end[ub - 1] - lo);
            unsigned long long cap = 2ull * (unsigned long long)(ub - ua);
            if (cap < 2) cap = 2;
            while (((span - 1ull) >> shift) + 1ull > cap) ++shift;               // points lo .. lo + span - 1
        }
    }
    cm[2 * c] = make_int4((int)lo, (int)(uint32_t)span, (int)(uint32_t)(span >> 32), shift);
    cm[2 * c + 1] = make_int4(2 * ua + 2 * c, ua, ub, 0);
}

__global__ void k_sub_records(const int4* __restrict__ cm, int32_t n_contigs, int64_t n_slots, const long long* __restrict__ u_start,
                              const long long* __restrict__ u_end, int4* __restrict__ rec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    int lo = 0, hi = n_contigs;
    while (lo < hi) { const int m = (lo + hi) >> 1; if ((int64_t)cm[2 * m + 1].x <= i) lo = m + 1; else hi = m; }
    const int c = lo - 1;
    int4 r = make_int4(0, -1, 0xffff | (4 << 16), 0);
    if (c >= 0) {
        const int4 m0 = cm[2 * c], m1 = cm[2 * c + 1];
        const int ua = m1.y, ub = m1.z, shift = m0.w;
        const unsigned long long span = (unsigned long long)(uint32_t)m0.y | ((unsigned long long)(uint32_t)m0.z << 32);
        const unsigned long long off0 = (unsigned long long)(i - (int64_t)m1.x) << shift;
        if (ub > ua && off0 < span) {
            const long long x0 = (long long)m0.x + (long long)off0;
            int k0; bool inside;
            sub_search(u_start, u_end, ua, ub, x0, k0, inside);
            uint32_t flags = inside ? 1u : 0u, t[3] = {0xffffu, 0xffffu, 0xffffu};
            if (shift > 15) flags |= 4u;                                         // offsets < 2^15: the 0xffff "none" mark stays above every point offset
            else {
                const long long W = 1ll << shift;
                int nt = 0;
                for (int k = k0; k < ub && nt < 4; ++k) {
                    if (!(k == k0 && inside)) {
                        if (u_start[k] - x0 >= W) break;
                        if (nt < 3) t[nt] = (uint32_t)(u_start[k] - x0);
                        ++nt;
                    }
                    if (u_end[k] - x0 >= W) break;
                    if (nt < 3) t[nt] = (uint32_t)(u_end[k] - x0);
                    ++nt;
                }
                if (nt > 3) flags |= 2u;
            }
            r = make_int4(k0, (int)(t[0] | (t[1] << 16)), (int)(t[2] | (flags << 16)), 0);
        }
    }
    rec[i] = r;
}

// K and IN of one point from its record: walk the (at most three) toggles at or below the point
__device__ __forceinline__ void sub_eval(const int4& r, uint32_t d, const long long* __restrict__ u_start, const long long* __restrict__ u_end,
                                         int ua, int ub, long long x, int& K, bool& in) {
    const uint32_t w1 = (uint32_t)r.y, w2 = (uint32_t)r.z, fl = w2 >> 16;
    const uint32_t t1 = w1 & 0xffffu, t2 = w1 >> 16, t3 = w2 & 0xffffu;
    if ((fl & 4u) || ((fl & 2u) && d >= t3)) { sub_search(u_start, u_end, r.x > ua ? r.x : ua, ub, x, K, in); return; }
    // toggles alternate end / start beginning with "end" when x0 is covered; an "end" toggle at or below the point moves K on
    const int n_le = (t1 <= d ? 1 : 0) + (t2 <= d ? 1 : 0) + (t3 <= d ? 1 : 0);   // "none" = 0xffff > d: inline offsets only for bins <= 2^15 wide
    const bool in0 = (fl & 1u) != 0;
    const int ends = in0 ? (n_le + 1) / 2 : n_le / 2;
    K = r.x + ends;
    in = in0 ^ ((n_le & 1) != 0);
}

// MODE 0: piece count per row; MODE 1: the pieces, at off[row]
template <bool STRICT, int MODE>
__global__ __launch_bounds__(PROBE_THREADS) void k_subtract_grid(SubGrid g, int32_t n_contigs, const long long* __restrict__ u_start,
                                                                 const long long* __restrict__ u_end, const int32_t* __restrict__ lc,
                                                                 const int32_t* __restrict__ lstart, const int32_t* __restrict__ lend,
                                                                 const int32_t* __restrict__ row_id, int64_t n, long long* __restrict__ cnt,
                                                                 const long long* __restrict__ off, int32_t* __restrict__ o_row,
                                                                 int32_t* __restrict__ o_start, int32_t* __restrict__ o_end) {
    const int64_t i = (int64_t)blockIdx.x * PROBE_THREADS + threadIdx.x;
    if (i >= n) return;
    const int32_t c = lc[i];
    const long long ls = lstart[i], le = (long long)lend[i] + (STRICT ? 0 : 1);
    long long pieces = 0;
    int first = 0, last = 0;
    bool in_s = false, in_y = false;
    if (le > ls) {
        pieces = 1;                                                              // nothing of the union on this contig: the row stays whole
        if ((uint32_t)c < (uint32_t)n_contigs) {
            const int4 m0 = g.cm[2 * c], m1 = g.cm[2 * c + 1];
            const int ua = m1.y, ub = m1.z;
            if (ub > ua) {
                const long long lo = m0.x;
                const unsigned long long span = (unsigned long long)(uint32_t)m0.y | ((unsigned long long)(uint32_t)m0.z << 32);
                // (scalars, not two-element arrays: indexed locals of this kernel went to 48 bytes of scratch per lane)
                const long long pt0 = ls, pt1 = le - 1;
                const int st0 = pt0 < lo ? 0 : ((unsigned long long)(pt0 - lo) >= span ? 1 : 2);
                const int st1 = pt1 < lo ? 0 : ((unsigned long long)(pt1 - lo) >= span ? 1 : 2);
                const uint32_t o0 = (uint32_t)(pt0 - lo), o1 = (uint32_t)(pt1 - lo);
                const uint32_t slot0 = (uint32_t)m1.x + (o0 >> m0.w), slot1 = (uint32_t)m1.x + (o1 >> m0.w);
                const uint32_t d0 = o0 & ((1u << m0.w) - 1u), d1 = o1 & ((1u << m0.w) - 1u);
                int4 r0 = make_int4(0, 0, 0, 0), r1 = make_int4(0, 0, 0, 0);
                if (st1 == 2) r1 = g.rec[slot1];
                if (st0 == 2) r0 = (st1 == 2 && slot0 == slot1) ? r1 : g.rec[slot0];
                int K0 = ua, K1 = ua;
                bool IN0 = false, IN1 = false;
                if (st0 == 1) K0 = ub;
                else if (st0 == 2) sub_eval(r0, d0, u_start, u_end, ua, ub, pt0, K0, IN0);
                if (st1 == 1) K1 = ub;
                else if (st1 == 2) sub_eval(r1, d1, u_start, u_end, ua, ub, pt1, K1, IN1);
                first = K0; in_s = IN0; in_y = IN1;
                last = K1 + (in_y ? 1 : 0);
                if (last > first) pieces = (in_s ? 0 : 1) + (long long)(last - first - 1) + (in_y ? 0 : 1);
            }
        }
    }
    if (MODE == 0) { cnt[i] = pieces; return; }
    if (pieces == 0) return;
    long long o = off[i];
    const int32_t rr = row_id ? row_id[i] : (int32_t)i;
    auto emit = [&](long long s, long long e) {
        o_row[o] = rr; o_start[o] = (int32_t)s; o_end[o] = (int32_t)(e - (STRICT ? 0 : 1)); ++o;
    };
    if (last <= first) { emit(ls, le); return; }
    if (!in_s) emit(ls, u_start[first]);
    for (int j = first; j + 1 < last; ++j) emit(u_end[j], u_start[j + 1]);
    if (!in_y) emit(u_end[last - 1], le);
}

}  // namespace ivj
