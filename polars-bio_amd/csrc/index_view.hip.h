// index_view.hip.h -- the build index as the kernels see it: layout, predicate, bound searches, window scans.
//
// HBM layout of the build index (sorted by (contig, start, row)):
//   b_start[Nb]  int32   start, the array the hi-bound search runs on
//   ep[Nb]       int2    (end, prefix-max of end inside the contig segment)
//   b_row[Nb]    int32   original build row
//   b_contig[Nb] int32   contig id in that order (rows outside the dictionary get n_contigs)
//   seg[n_contigs + 2]   segment offsets; seg[n_contigs] = number of valid rows
//   e_end / e_pos [Nb]   optional: ends sorted by (contig, end, position), and that position
//   cmeta[n_contigs]     per-contig {segment, min/max start, bin shift, table offset}
//   rec4[Nb]     int4    {start, end, row, pmax}: the one read per candidate of the flat overlap path (flat.hip.h)
//   tab2[2 Nb + 2 n_contigs] uint2 per start bin: {first position of the bin, first position whose prefix max
//                        reaches the bin's lower edge}
//   brec[2 Nb + 2 n_contigs] direct-address table over start, 16 B per bin: first position of the bin (bit 31: the bin
//                        holds more rows than the record shows) and the keys of its first SIX rows as 16-bit offsets
//                        from the bin's lower edge (0xffff: no such row) -- bins wider than 2^16: the 32-bit keys of
//                        rows p0 .. p0+2 instead; about one build row per bin, so the hi-bound of a probe is ONE
//                        16-byte gather instead of a log2(Nb)-step binary search of dependent gathers, and a wavefront
//                        practically never waits for a lane that has to search a crowded bin
//
// Predicate (polars_bio/range_op.py:75-84; src/option.rs:95-100):
//   STRICT: q.start <  b.end && b.start <  q.end      WEAK: <=
// For a probe q on contig c with segment [a,b):
//   hi = first p in [a,b) with !(b_start[p] (<) q.end)     -> every match has p < hi
//   matches = { p in [a,hi) : q.start (<) end[p] };  the prefix max bounds the backward scan:
//   stop at the first p (going down) with !(q.start (<) pmax[p]).
#pragma once
#include "scan.hip.h"

namespace ivj {

constexpr int PROBE_THREADS = 256;
constexpr int PROBE_ITEMS = 4;   // probes per thread of the overlap count / fill / fused kernels
constexpr int PROBE_ITEMS_LAT = 2;   // nearest and the dense fill: shorter per-thread chains, full occupancy
constexpr int PROBE_TILE = PROBE_THREADS * PROBE_ITEMS;
constexpr int CM_LDS = 256;      // per-contig grid metadata is copied to LDS by the per-probe kernels up to this many contigs

constexpr int HIER_MAX = 8;                             // levels of the block maxima incl. level 0: 16^7 rows
struct HierView {
    const int32_t* v;
    int nlev;                                           // highest level
    uint32_t off[HIER_MAX];                             // offset of level l in v
};

struct IndexView {
    const int32_t* b_start;
    const int2* ep;
    const int32_t* b_row;
    const int32_t* seg;
    const int32_t* e_end;
    const int32_t* e_pos;
    const int32_t* flags;  // flags[0] != 0: some build row has start > end
    const int4* cmeta;     // per contig: {a, b, ulo, uhi} {shift, tb, 0, 0}  (two int4)
    const int4* brec;      // direct-address table: brec[tb + j] = {p0 | more << 31, six 16-bit key offsets} (bins <= 2^16 wide)
                           // or {p0 | more << 31, key[p0], key[p0+1], key[p0+2]}; p0 = first position whose ustart >= ulo + (j << shift)
    const uint32_t* bins;  // the same table as plain first positions (4 B per bin): used instead of brec for
                           // small build sides, whose 4-byte tables + key arrays stay L2-resident
    const int4* cmeta_e;   // the same pair of structures over the end-sorted order (e_end)
    const int4* brec_e;
    const uint32_t* bins_e;
    int32_t use_rec;       // 1: gather 16-byte records, 0: 4-byte bins + bound search on the key array
    const int32_t* pargmax; // position of the first row that attains ep[p].y (prefix max) -- nearest only
    const int4* cmeta_j;    // count_overlaps: ONE bin grid per contig shared by the start- and the end-sorted order
    const int4* crec;       //   crec[slot] = {first start position | more << 31, first end position | more << 31, 2 x 16-bit start offsets, 2 x 16-bit end offsets}
    const int4* nline;      // nearest, k = 1: ONE 128-byte line per start-table slot = {brec, nrec of the slot's first three positions} (k_nearest_lines)
    const int4* nrec;       // nearest: nrec[2p] = {pmax[p-1], row of its argmax, start[p], end[p]}, nrec[2p+1] = {row of p, v1, row1, v2}
                            // (left / right candidate of hi = p; the two prefix-max levels below the current one)
    const int4* rec4;       // flat overlap path: {start, end, build row, prefix max} per sorted position
    const uint2* tab2;      //   tab2[slot] = {first position of start bin `slot`, first position whose prefix max reaches its lower edge}
    int32_t n_contigs;
    HierView hier;          // the sorted ends and their block maxima: windows that run on (hier_walk)
};

__device__ __forceinline__ uint32_t flip(int32_t v) { return (uint32_t)v ^ 0x80000000u; }
__device__ __forceinline__ int32_t unflip(uint32_t v) { return (int32_t)(v ^ 0x80000000u); }

template <bool STRICT>
__device__ __forceinline__ bool lt_op(int32_t x, int32_t y) { return STRICT ? (x < y) : (x <= y); }

// gfx950 needs two wait states between a VALU write of an SGPR pair / VCC and a VALU that reads it as a carry or lane mask
// (the compiler pads every v_cmp -> v_cndmask / v_addc pair with s_nop 1).  The two hot compare sequences of the join are
// therefore written out with rotating SGPR pairs, every consumer three instructions behind its compare: no padding.
//
// Mask of sixteen ends: bit i <=> qs (<) e[i] for the sixteen ends e[0..15]; row 15 is shifted in first (m = m + m + carry).
#define IVJ_ENDS_MASK_ASM(CMP)                                                                                                     \
    asm("v_cmp_" CMP "_i32_e64 %1, %5, %21\n\tv_cmp_" CMP "_i32_e64 %2, %5, %20\n\tv_cmp_" CMP "_i32_e64 %3, %5, %19\n\t"     \
        "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\tv_cmp_" CMP "_i32_e64 %1, %5, %18\n\t"                                       \
        "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\tv_cmp_" CMP "_i32_e64 %2, %5, %17\n\t"                                       \
        "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\tv_cmp_" CMP "_i32_e64 %3, %5, %16\n\t"                                       \
        "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\tv_cmp_" CMP "_i32_e64 %1, %5, %15\n\t"                                       \
        "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\tv_cmp_" CMP "_i32_e64 %2, %5, %14\n\t"                                       \
        "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\tv_cmp_" CMP "_i32_e64 %3, %5, %13\n\t"                                       \
        "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\tv_cmp_" CMP "_i32_e64 %1, %5, %12\n\t"                                       \
        "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\tv_cmp_" CMP "_i32_e64 %2, %5, %11\n\t"                                       \
        "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\tv_cmp_" CMP "_i32_e64 %3, %5, %10\n\t"                                       \
        "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\tv_cmp_" CMP "_i32_e64 %1, %5, %9\n\t"                                        \
        "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\tv_cmp_" CMP "_i32_e64 %2, %5, %8\n\t"                                        \
        "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\tv_cmp_" CMP "_i32_e64 %3, %5, %7\n\t"                                        \
        "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\tv_cmp_" CMP "_i32_e64 %1, %5, %6\n\t"                                        \
        "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\t"                                                                             \
        "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\t"                                                                             \
        "v_addc_co_u32_e64 %0, %4, %0, %0, %1"                                                                                  \
        : "+v"(m), "=&s"(ta), "=&s"(tb), "=&s"(tc), "=&s"(td)                                                                   \
        : "v"(qs), "v"(v0.x), "v"(v0.y), "v"(v0.z), "v"(v0.w), "v"(v1.x), "v"(v1.y), "v"(v1.z), "v"(v1.w), "v"(v2.x), "v"(v2.y),  \
          "v"(v2.z), "v"(v2.w), "v"(v3.x), "v"(v3.y), "v"(v3.z), "v"(v3.w))
template <bool STRICT>
__device__ __forceinline__ uint32_t ends_mask16(int32_t qs, const int4& v0, const int4& v1, const int4& v2, const int4& v3) {
    uint32_t m = 0;
    unsigned long long ta, tb, tc, td;
    if (STRICT) IVJ_ENDS_MASK_ASM("lt");
    else IVJ_ENDS_MASK_ASM("le");
    return m;
}

// ---- the sorted ends and their block maxima ("hier"): windows that run on --------------------------------------------------------
// A probe whose window is not settled by the few rows below its hi-bound (the prefix max there is still above its start) has to
// find every row further down that ends above its start.  With a tail of long intervals in the build side (genes among exons, a
// contig-wide row) the prefix max stays high for hundreds to millions of rows, of which a handful match: a row-by-row scan is what
// makes sorted-window joins fall off a cliff on such inputs.  hier level 0 = the ends in sorted order (a compact copy: one
// 64-byte line = one block of sixteen), level l >= 1, entry i = max end over the sorted rows [i << 4l, (i + 1) << 4l) (blocks
// may straddle contigs, the walk stops at the contig's first row).  hier_walk enters a block only when its maximum is above the
// probe's start, i.e. only blocks that hold a match, and reads a block with four 16-byte loads: a few dependent loads per match,
// whatever the intervals look like.  (1 + 1/15) n values per index, every level padded to whole blocks.
struct HierShape {
    int nlev;
    uint32_t off[HIER_MAX];
    int64_t len[HIER_MAX];
    size_t values;                                      // allocation, in values
};
inline HierShape hier_shape(int64_t n) {
    HierShape h;
    h.nlev = 0;
    int64_t o = 0, len = n;
    for (int l = 0; l < HIER_MAX; ++l) {
        h.off[l] = (uint32_t)o; h.len[l] = 0;
        if (l == 0 || len > 16) {
            if (l > 0) len = (len + 15) / 16;
            h.len[l] = len; h.nlev = l;
            o += (len + 15) & ~(int64_t)15;
        } else len = 0;
    }
    h.values = (size_t)o + 16;
    return h;
}
// Levels 0 .. 3 in one pass over ep: one workgroup per 4096 rows = ONE level-3 entry (256 threads x 16 rows; level 1 = the
// thread's maximum, level 2 = the maximum of sixteen neighbouring threads, level 3 = the workgroup's); pads of the last blocks =
// INT32_MIN.  The levels above (a thousand values at most) are a second, one-workgroup launch.
constexpr int HIER_WG_ROWS = 4096;
__global__ __launch_bounds__(256) void k_hier_low(const int2* __restrict__ ep, int64_t n, int32_t* __restrict__ v, HierShape h) {
    __shared__ int32_t s_w[256 / kWave];
    const int64_t row0 = (int64_t)blockIdx.x * HIER_WG_ROWS + (int64_t)threadIdx.x * 16;
    int32_t m = INT32_MIN;
    int32_t e[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        e[t] = row0 + t < n ? ep[row0 + t].x : INT32_MIN;
        m = e[t] > m ? e[t] : m;
    }
    const int64_t pad0 = (n + 15) & ~(int64_t)15;
    if (row0 < pad0) {
        int4* o = reinterpret_cast<int4*>(v + row0);
        o[0] = make_int4(e[0], e[1], e[2], e[3]); o[1] = make_int4(e[4], e[5], e[6], e[7]);
        o[2] = make_int4(e[8], e[9], e[10], e[11]); o[3] = make_int4(e[12], e[13], e[14], e[15]);
    }
    const int64_t i1 = row0 >> 4;
    if (h.nlev >= 1 && i1 < ((h.len[1] + 15) & ~(int64_t)15)) v[h.off[1] + i1] = m;
    int32_t m2 = m;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) { const int32_t o = __shfl_xor(m2, d, kWave); m2 = o > m2 ? o : m2; }
    const int64_t i2 = row0 >> 8;
    if (h.nlev >= 2 && (threadIdx.x & 15) == 0 && i2 < ((h.len[2] + 15) & ~(int64_t)15)) v[h.off[2] + i2] = m2;
    if (h.nlev < 3) return;                                                    // uniform
    int32_t m3 = m2;
#pragma unroll
    for (int d = 16; d < kWave; d <<= 1) { const int32_t o = __shfl_xor(m3, d, kWave); m3 = o > m3 ? o : m3; }
    if ((threadIdx.x & (kWave - 1)) == 0) s_w[threadIdx.x / kWave] = m3;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256 / kWave; ++w) m3 = s_w[w] > m3 ? s_w[w] : m3;
        if ((int64_t)blockIdx.x < ((h.len[3] + 15) & ~(int64_t)15)) v[h.off[3] + blockIdx.x] = m3;
    }
}
// Levels 4 and up from level 3 (and the pads of level 3's last block): one workgroup, level after level.
__global__ __launch_bounds__(256) void k_hier_high(int32_t* __restrict__ v, HierShape h, int64_t low_wgs) {
    const int64_t pad3 = (h.len[3] + 15) & ~(int64_t)15;
    for (int64_t i = low_wgs + threadIdx.x; i < pad3; i += blockDim.x) v[h.off[3] + i] = INT32_MIN;
    __syncthreads();
    for (int l = 4; l <= h.nlev; ++l) {
        const int32_t* src = v + h.off[l - 1];
        int32_t* dst = v + h.off[l];
        const int64_t n_src = h.len[l - 1], padded = (h.len[l] + 15) & ~(int64_t)15;
        for (int64_t i = threadIdx.x; i < padded; i += blockDim.x) {
            int32_t mx = INT32_MIN;
            for (int t = 0; t < 16; ++t) {
                if (i * 16 + t < n_src) { const int32_t x = src[i * 16 + t]; mx = x > mx ? x : mx; }
            }
            dst[i] = mx;
        }
        __syncthreads();
    }
}

// Rows at or below sorted position i (down to seg_a, the contig's first row) that end above qsv, in descending position; every
// one of them must start below the probe's end (i < hi-bound).  ep_at(p) = {end, prefix max} of row p; f(p) returns false to stop.  Depth-first over the
// block maxima, right to left, one 16-entry block per step: the entries at or left of the cursor that are above qsv are rows to
// report (level 0) or the child to enter (the rightmost one); an exhausted block hands over to the entries left of its parent,
// and the prefix max of the row below the subtree just left says whether anything further down can still match.
template <bool STRICT, class EpAt, class F>
__device__ __forceinline__ void hier_walk(const HierView& H, const EpAt& ep_at, int seg_a, int i, int32_t qsv, F&& f) {
    int lv = 0;
    int chk = -1;                                                              // row whose prefix max is still to be looked at
    while (i >= 0) {
        if ((((int64_t)(i + 1) << (4 * lv)) - 1) < (int64_t)seg_a) return;     // the entry lies below the contig
        const int base = i & ~15;
        uint32_t off = 0;
#pragma unroll
        for (int l = 1; l < HIER_MAX; ++l) off = lv == l ? H.off[l] : off;
        const int4* bp = reinterpret_cast<const int4*>(H.v + (size_t)off + (size_t)base);
        const int4 w0 = bp[0], w1 = bp[1], w2 = bp[2], w3 = bp[3];
        if (chk >= 0) {
            if (!lt_op<STRICT>(qsv, ep_at(chk).y)) return;                     // nothing at or below row chk reaches the probe
            chk = -1;
        }
        uint32_t m = ends_mask16<STRICT>(qsv, w0, w1, w2, w3) & ((2u << (i & 15)) - 1u);
        if (lv == 0) {
            if (seg_a > base) m &= ~((1u << (seg_a - base)) - 1u);
            while (m) { const int j = 31 - __builtin_clz(m); m ^= 1u << j; if (!f(base + j)) return; }
        } else if (m) {
            i = ((base + 31 - __builtin_clz(m)) << 4) + 15;                    // the rightmost child above qsv, all of it
            --lv;
            continue;
        }
        // block exhausted: the entries left of its parent (of the first ancestor that has any)
        int node = base;
        do {
            if (lv == H.nlev) return;
            node >>= 4; ++lv;
        } while ((node & 15) == 0);
        const int64_t b = (int64_t)node << (4 * lv);                           // first row of the subtree just left
        if (b <= (int64_t)seg_a) return;
        chk = (int)(b - 1);
        i = node - 1;
    }
}

// The same enumeration in ASCENDING position over the rows [lo, hi) (all of one contig): rows that end above qsv, smallest
// (start, row) first -- the order nearest(k > 1) lists overlapping rows in.  Cursor = an entry none of whose rows has been looked
// at; a block hands over to the entry right of its parent.  f(p) returns false to stop.
template <bool STRICT, class F>
__device__ __forceinline__ void hier_walk_up(const HierView& H, int lo, int hi, int32_t qsv, F&& f) {
    int i = lo, lv = 0;
    while (((int64_t)i << (4 * lv)) < (int64_t)hi) {
        const int base = i & ~15;
        uint32_t off = 0;
#pragma unroll
        for (int l = 1; l < HIER_MAX; ++l) off = lv == l ? H.off[l] : off;
        const int4* bp = reinterpret_cast<const int4*>(H.v + (size_t)off + (size_t)base);
        const int4 w0 = bp[0], w1 = bp[1], w2 = bp[2], w3 = bp[3];
        uint32_t m = ends_mask16<STRICT>(qsv, w0, w1, w2, w3) & ~((1u << (i & 15)) - 1u);
        if (lv == 0) {
            if (hi - base < 16) m &= (1u << (hi - base)) - 1u;
            while (m) { const int j = __builtin_ctz(m); m &= m - 1; if (!f(base + j)) return; }
        } else if (m) {
            i = (base + __builtin_ctz(m)) << 4;                                // the leftmost child above qsv, from its first entry
            --lv;
            continue;
        }
        if (lv == H.nlev) return;
        i = (base >> 4) + 1; ++lv;
        while ((i & 15) == 0 && lv < H.nlev) { i >>= 4; ++lv; }
    }
}

// Workgroup b is observed to run on XCD b % 8.  Giving every XCD one CONTIGUOUS eighth of the (bucket-ordered) tiles
// means its L2 only ever holds the index slices of its own buckets, instead of all eight L2s fetching every slice.
// Launch 8 * ceil(ntiles / 8) workgroups; a tile index >= ntiles has nothing to do.  Speed only, never the result.
__device__ __forceinline__ long long xcd_tile64(long long block, long long ntiles) {
    const long long per = (ntiles + 7) / 8;
    return (block & 7) * per + (block >> 3);
}

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
            // the record answers the rank inside the bin without touching the key array: six inline 16-bit key offsets
            // (narrow bins), or the keys of rows p0 .. p0+2 (bins wider than 2^16).  Keys of later bins are >= the bin's
            // upper edge > target, so only rows of this bin count; "more" = the bin holds rows the record does not show.
            const int p0 = rec[k].x & 0x7fffffff;
            const bool more = rec[k].x < 0;
            int lo;
            bool full;
            if (m1[k].x <= 16) {
                const uint32_t toff = ((uint32_t)tu[k] - (uint32_t)m0[k].z) & ((1u << m1[k].x) - 1u);
                const uint32_t w1 = (uint32_t)rec[k].y, w2 = (uint32_t)rec[k].z, w3 = (uint32_t)rec[k].w;
                const int cnt = ((w1 & 0xffffu) < toff ? 1 : 0) + ((w1 >> 16) < toff ? 1 : 0) + ((w2 & 0xffffu) < toff ? 1 : 0) +
                                ((w2 >> 16) < toff ? 1 : 0) + ((w3 & 0xffffu) < toff ? 1 : 0) + ((w3 >> 16) < toff ? 1 : 0);
                lo = p0 + cnt;
                full = cnt == 6;
            } else {
                const bool n0 = (unsigned long long)flip(rec[k].y) < tu[k];
                const bool n1 = n0 && (unsigned long long)flip(rec[k].z) < tu[k];
                const bool n2 = n1 && (unsigned long long)flip(rec[k].w) < tu[k];
                lo = p0 + (n0 ? 1 : 0) + (n1 ? 1 : 0) + (n2 ? 1 : 0);
                full = n2;
            }
            if (full && more) {
                // crowded bin: gallop up the key array from the rows already counted (bounded by the contig segment; the
                // rows of the bin end before the first key >= target), then a bound search
                int step = 1;
                while (lo + step - 1 < b[k] && (unsigned long long)flip(keys[lo + step - 1]) < tu[k]) { lo += step; step <<= 1; }
                int hi = lo + step - 1 < b[k] ? lo + step - 1 : b[k];
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
// Streaming variant: the probe columns are read exactly once, so they are fetched with the non-temporal hint and
// do not push the (re-used) lookup tables out of the XCD's L2.
template <int N>
__device__ __forceinline__ void load_items_nt(const int32_t* __restrict__ p, int64_t i0, int64_t n, bool vec_ok,
                                              int32_t fill, int32_t (&out)[N]) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v2i __attribute__((ext_vector_type(2)));
    if (vec_ok && i0 + N <= n) {
        if constexpr (N == 4) {
            const v4i v = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(p + i0));
            out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
            return;
        } else if constexpr (N == 2) {
            const v2i v = __builtin_nontemporal_load(reinterpret_cast<const v2i*>(p + i0));
            out[0] = v.x; out[1] = v.y;
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = (i0 + k < n) ? __builtin_nontemporal_load(p + i0 + k) : fill;
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

}  // namespace ivj
