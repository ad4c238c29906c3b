// cslice.hip.h -- pb.overlap, fused single pass on CONTIG-ALIGNED index slices with 12-byte probe records (round 3).
//
// Same idea as slice.hip.h (probes partitioned by the slice of the sorted build side that holds their hi-bound, the
// slice resident in LDS), rebuilt around what round 2's counters said about k_slice_join: it was not bandwidth-bound
// but issue-bound -- 587 VALU + 392 SALU instructions per 64 probes, 60 % of a wavefront's life parked behind
// dependent LDS reads in divergent loops.  This path removes work instead of hiding it:
//
//   * slices never cross a contig (k_cs_prep cuts every contig's segment into ceil(n_c / R) slices), so a bucket's
//     probes all carry the slice's contig: the record is {start, end, row} = 12 bytes (-25 % scatter writes and join
//     reads) and the join needs no per-probe contig test, segment lookup or row-limit arithmetic;
//   * per-slice start bins are built ONCE per index (k_cs_bins) instead of by each of the ~7 workgroups that visit a slice;
//   * hi-bound = one 2-byte bin read + four start compares, branch-free (a bin holds 0.5 rows on average; more than four
//     rows below the probe's end inside its bin take a rare fallback);
//   * window = sixteen ends as four ds_read_b128 + sixteen v_cmp / v_addc pairs that shift the match bits into a mask
//     (two VALU instructions per row), one prefix-max read decides whether the window may run on;
//   * emission stages ONE 4-byte entry {wave-local probe slot, slice-local row} per pair (no LDS read inside the
//     per-lane bit loop); probe row and build row are resolved at copy-out, where the LDS reads are independent;
//   * wavefront scans use DPP adds (no ds_bpermute), tiles are 4096 probes (24 k output reservations instead of 49 k
//     on the one cursor), the tile loop is barrier-free as in slice.hip.h (split-phase reservation through LDS control
//     blocks).
//
// Restrictions (the callers fall back to slice.hip.h otherwise): dictionaries of up to CS_MAX_CONTIGS contigs, at most
// SL_MAX_BUCKETS slices.  Output order: as slice.hip.h's fused mode -- the pairs of one probe row are contiguous and
// ordered by (build.start, build row); tiles land in reservation order.
#pragma once
#include "slice.hip.h"

namespace ivj {

constexpr int CS_THREADS = 1024;
constexpr int CS_WAVES = CS_THREADS / kWave;
constexpr int CS_ITEMS = 4;
constexpr int CS_TILE = CS_THREADS * CS_ITEMS;          // probes per tile of the partition and of the join
constexpr int CS_WTILE = kWave * CS_ITEMS;              // probes of one wavefront per tile (its private slot range)
constexpr int CS_MAX_CONTIGS = 256;
constexpr int CS_WIN = 12;                              // rows below hi the branch-free window is guaranteed to cover
constexpr int CS_LIN = 4;                              // rows below the window a running-on probe checks one by one (LDS) before it takes the block maxima
constexpr int CS_POS_BIAS = 1 << 23;                    // staging entries carry (row - first row of the slice) + bias in 24 bits: rows below the slice too
static_assert((long long)SL_MAX_BUCKETS * SL_MAX_ROWS <= CS_POS_BIAS, "an index of this path must fit a staging entry's row field");
constexpr int CS_BIN_STRIDE_PAD = 2;                    // bins per slice in global memory: 2 R + 2 (u16)

constexpr int CS_CUR_STRIDE = 1;                        // words between two region cursors (32 = one cursor per 128-byte line: measured, no change -- the
                                                        // latency of the returning atomics, 8 - 11 us under load, does not come from lines shared by cursors)
constexpr int CS_META_FMT = 8;                          // meta[8]: record format of the call (k_cs_regions): 0 = 12-byte records, LB > 0 = 8-byte records
constexpr unsigned long long CS_STATE_REC8 = 8ull;     // state word, value 8 (bit 3): a probe did not fit the 8-byte record form
constexpr int CS_REC8_GIVE_UP = 3;                      // overflows in a row after which a context stops offering the 8-byte form (host_cslice.hip.h)
__host__ __device__ __forceinline__ int cs_bits_for(uint32_t v) { int b = 0; while (b < 32 && (v >> b) != 0) ++b; return b == 0 ? 1 : b; }

typedef int cs_rec __attribute__((ext_vector_type(3), aligned(4)));      // one probe record {start, end, row}: 12 bytes, 4-byte aligned
typedef int cs_rec8 __attribute__((ext_vector_type(2), aligned(8)));     // round 5, where a call's probes fit: {(end - slice minimum) << LB | (end - start), row}

struct CsGeom {
    int nb;            // bucket SLOTS = upper bound on the number of slices (host-known); bucket nb = probes without a candidate row
    int R;             // rows per slice (multiple of 64); a contig's last slice may be shorter
    int ncells;        // cells of the direct-address bucket table (upper bound, host-known)
    int cps;           // cells per slice
    int n_contigs;
};

// per-slice metadata written by k_cs_bins (two int4 per slice)
//   [0] = {min start, bin shift, number of cells, prefix max of the row below the slice (INT32_MIN: none)}
//   [1] = {first row of the contig's segment, contig, rows, first row}

__device__ __forceinline__ int wave_incl_sum_dpp(int v) {
    // row_shr:1,2,4,8 inside the rows of 16 lanes, then row_bcast:15 / row_bcast:31 across the rows
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

// (the two-wait-state rule and the sixteen-end window mask, ends_mask16, live in index_view.hip.h)
// h + number of the four starts below the probe's end (s (<) qe); *m4 = lanes whose fourth start is still below it
#define IVJ_CS_COUNT4_ASM(CMP)                                                                                                     \
    asm("v_cmp_" CMP "_i32_e64 %1, %6, %10\n\tv_cmp_" CMP "_i32_e64 %2, %7, %10\n\tv_cmp_" CMP "_i32_e64 %3, %8, %10\n\t"      \
        "v_cmp_" CMP "_i32_e64 %4, %9, %10\n\t"                                                                                \
        "v_addc_co_u32_e64 %0, %5, %0, 0, %1\n\tv_addc_co_u32_e64 %0, %5, %0, 0, %2\n\t"                                      \
        "v_addc_co_u32_e64 %0, %5, %0, 0, %3\n\tv_addc_co_u32_e64 %0, %5, %0, 0, %4"                                           \
        : "+v"(h), "=&s"(ta), "=&s"(tb), "=&s"(tc), "=&s"(td), "=&s"(te)                                                        \
        : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(qe))
template <bool STRICT>
__device__ __forceinline__ int cs_count4(int h, int32_t s0, int32_t s1, int32_t s2, int32_t s3, int32_t qe, unsigned long long* m4) {
    unsigned long long ta, tb, tc, td, te;
    if (STRICT) IVJ_CS_COUNT4_ASM("lt");
    else IVJ_CS_COUNT4_ASM("le");
    *m4 = td;
    return h;
}

// ---- slices, splitters, bucket table: ONE workgroup -------------------------------------------------------------------------
// Contig c's segment [seg[c], seg[c+1]) is cut into ns_c = ceil(n_c / R) slices; fs_c = slices of the contigs before it.
//   bound[j]  first sorted row of slice j (j >= number of slices: the number of valid rows), nb + 1 entries
//   spl[j]    composite (contig, start) key of that row (~0 for the unused slots)
//   cm[c]     {ulo, uhi, shift, first cell | fs_c << 16}: uniform grid over the starts of the contig's rows
//   cell[i]   lo | hi << 16: range of "number of splitters below" a key of that cell can have
__global__ __launch_bounds__(CS_THREADS) void k_cs_prep(const int32_t* __restrict__ seg, const int32_t* __restrict__ b_start, CsGeom g,
                                                       int32_t* __restrict__ bound, unsigned long long* __restrict__ spl,
                                                       int4* __restrict__ cm, uint32_t* __restrict__ cell,
                                                       uint32_t* __restrict__ zero_a, int n_zero_a, uint32_t* __restrict__ zero_b, int n_zero_b) {
    // (round 5: the call state and the sample histogram of a call that launches this kernel are cleared here -- two fill
    // operations less in the stream; every consumer of the words is queued behind this kernel)
    for (int i = threadIdx.x; i < n_zero_a; i += CS_THREADS) zero_a[i] = 0u;
    for (int i = threadIdx.x; i < n_zero_b; i += CS_THREADS) zero_b[i] = 0u;
    __shared__ int4 l_cm[CS_MAX_CONTIGS];
    __shared__ int l_a[CS_MAX_CONTIGS + 1], l_fs[CS_MAX_CONTIGS + 1];
    __shared__ uint32_t l_lo[CS_MAX_CONTIGS], l_hi[CS_MAX_CONTIGS];
    __shared__ unsigned long long l_spl[SL_MAX_BUCKETS + 1];
    __shared__ int l_cells;
    const int tid = threadIdx.x, nc = g.n_contigs;
    for (int c = tid; c <= nc; c += CS_THREADS) {
        const int a = seg[c];
        l_a[c] = a;
        if (c < nc) {
            const int b = seg[c + 1];
            l_lo[c] = b > a ? flip(b_start[a]) : 0u;
            l_hi[c] = b > a ? flip(b_start[b - 1]) : 0u;
        }
    }
    __syncthreads();
    // per contig, in parallel: slices, cells, cell shift (the shift loop is the long part: up to 31 trips of 64-bit shifts -- round 4 ran
    // it for every contig on ONE thread); then one thread lays the prefix sums down
    __shared__ int l_ns[CS_MAX_CONTIGS], l_ncl[CS_MAX_CONTIGS], l_sh[CS_MAX_CONTIGS];
    for (int c = tid; c < nc; c += CS_THREADS) {
        const int a = l_a[c], b = l_a[c + 1];
        const int ns = (b - a + g.R - 1) / g.R;
        int ncl = g.cps * ns;
        if (ncl < 2) ncl = 2;                                      // two cells keep the shift <= 31 for any int32 span
        const uint32_t ulo = l_lo[c], uhi = l_hi[c];
        int shift = 0;
        while (shift < 31 && ((unsigned long long)(uhi - ulo) >> shift) + 1ull > (unsigned long long)ncl) ++shift;
        l_ns[c] = ns; l_ncl[c] = ncl; l_sh[c] = shift;
    }
    __syncthreads();
    if (tid == 0) {
        int fs = 0, tb = 0;
        for (int c = 0; c < nc; ++c) {
            l_cm[c] = make_int4((int)l_lo[c], (int)l_hi[c], l_sh[c], tb | (fs << 16));
            l_fs[c] = fs;
            fs += l_ns[c]; tb += l_ncl[c];
        }
        l_fs[nc] = fs;
        l_cells = tb;
    }
    __syncthreads();
    const int total = l_fs[nc];                                    // number of slices (<= g.nb by construction of R)
    const int nvalid = l_a[nc];
    for (int c = tid; c < nc; c += CS_THREADS) cm[c] = l_cm[c];
    for (int j = tid; j <= g.nb; j += CS_THREADS) {
        int row = nvalid;
        unsigned long long key = ~0ull;
        if (j < total) {
            int lo = 0, hi = nc;                                   // last contig with fs <= j that owns slices
            while (lo < hi) { const int m = (lo + hi) >> 1; if (l_fs[m + 1] <= j) lo = m + 1; else hi = m; }
            row = l_a[lo] + (j - l_fs[lo]) * g.R;
            key = ((unsigned long long)(uint32_t)lo << 32) | (unsigned long long)flip(b_start[row]);
        }
        bound[j] = row;
        if (j < g.nb) { spl[j] = key; }
        l_spl[j < SL_MAX_BUCKETS ? j : SL_MAX_BUCKETS] = key;
    }
    __syncthreads();
    const int ncl_total = l_cells;
    for (int i = tid; i < ncl_total; i += CS_THREADS) {
        int lo = 0, hi = nc;                                       // contig of cell i: last one whose first cell is <= i
        while (lo < hi) { const int m = (lo + hi) >> 1; if ((l_cm[m].w & 0xffff) <= i) lo = m + 1; else hi = m; }
        const int c = lo - 1;
        const int4 m = l_cm[c];
        const int k = i - (m.w & 0xffff);
        const int jlo = l_fs[c], jhi = l_fs[c + 1];
        const int last = (c + 1 < nc ? (l_cm[c + 1].w & 0xffff) : ncl_total) - 1;
        auto below = [&](int kk) {                                 // splitters of the dictionary below the lower edge of cell kk
            if (kk <= 0) return jlo;
            const unsigned long long edge = ((unsigned long long)(uint32_t)c << 32) + (unsigned long long)(uint32_t)m.x +
                                            ((unsigned long long)kk << m.z);
            int a = jlo, b = jhi;
            while (a < b) { const int mid = (a + b) >> 1; if (l_spl[mid] < edge) a = mid + 1; else b = mid; }
            return a;
        };
        const int l = below(k);
        const int h = i == last ? jhi : below(k + 1);
        cell[i] = (uint32_t)l | ((uint32_t)h << 16);
    }
}

// Bucket of a probe (contig c, end qe): the slice of contig c that holds row hi - 1, hi = number of the contig's rows whose
// start lies below the end [Weak: at or below]; no such row (or a contig outside the dictionary / without rows): bucket g.nb.
template <bool STRICT>
__device__ __forceinline__ uint32_t cs_bucket(const unsigned long long* __restrict__ l_spl, const int4* __restrict__ l_cm,
                                              const uint32_t* __restrict__ l_cell, int nb, int32_t n_contigs, int32_t c, int32_t qe) {
    if ((uint32_t)c >= (uint32_t)n_contigs) return (uint32_t)nb;
    const unsigned long long tu = (unsigned long long)flip(qe) + (STRICT ? 0ull : 1ull);
    const unsigned long long key = ((unsigned long long)(uint32_t)c << 32) + tu;        // (c, INT32_MAX) + 1 carries into the contig
    const int4 m = l_cm[c];
    const uint32_t ulo = (uint32_t)m.x, uhi = (uint32_t)m.y;
    uint32_t k;
    if (tu <= ulo) k = 0;
    else if (tu > uhi) k = (uhi - ulo) >> m.z;
    else k = ((uint32_t)tu - ulo) >> m.z;
    const uint32_t lh = l_cell[(m.w & 0xffff) + k];
    int pos = (int)(lh & 0xffffu);
    const int hi = (int)(lh >> 16);
    while (pos < hi && l_spl[pos] < key) ++pos;
    return pos <= (int)((uint32_t)m.w >> 16) ? (uint32_t)nb : (uint32_t)(pos - 1);
}

// The same bucket without divergent control flow (round 6, the wide-tile scatter): the scatter's lookups ran ~ 5 exec-mask branches per
// probe (83 M scalar + 33 M branch instructions per launch next to 148 M vector ones).  A contig outside the dictionary is looked up as
// contig 0 and answered with nb at the end; the cell's (at most two, nearly always) splitters are compared at once; only a cell with more
// than two splitters below the key takes the loop (one uniform branch on a ballot).  l_spl is read up to two entries behind its end.
template <bool STRICT>
__device__ __forceinline__ uint32_t cs_bucket_flat(const unsigned long long* __restrict__ l_spl, const int4* __restrict__ l_cm,
                                                   const uint32_t* __restrict__ l_cell, int nb, int32_t n_contigs, int32_t c, int32_t qe) {
    const bool okc = (uint32_t)c < (uint32_t)n_contigs;
    const int cc = okc ? c : 0;
    const unsigned long long tu = (unsigned long long)flip(qe) + (STRICT ? 0ull : 1ull);
    const unsigned long long key = ((unsigned long long)(uint32_t)cc << 32) + tu;
    const int4 m = l_cm[cc];
    const uint32_t ulo = (uint32_t)m.x, uhi = (uint32_t)m.y;
    unsigned long long tc = tu < (unsigned long long)ulo ? (unsigned long long)ulo : tu;
    tc = tc > (unsigned long long)uhi ? (unsigned long long)uhi : tc;
    const uint32_t k = ((uint32_t)tc - ulo) >> m.z;
    const uint32_t lh = l_cell[(m.w & 0xffff) + k];
    int pos = (int)(lh & 0xffffu);
    const int hi = (int)(lh >> 16);
    const unsigned long long s0 = l_spl[pos], s1 = l_spl[pos + 1];
    const bool c0 = pos < hi && s0 < key;
    const bool c1 = c0 && pos + 1 < hi && s1 < key;
    pos += (int)c0 + (int)c1;
    if (__builtin_expect(__ballot(c1 && pos < hi) != 0ull, 0)) {                // uniform, rare
        if (c1) { while (pos < hi && l_spl[pos] < key) ++pos; }
    }
    const uint32_t b = pos <= (int)((uint32_t)m.w >> 16) ? (uint32_t)nb : (uint32_t)(pos - 1);
    return okc ? b : (uint32_t)nb;
}

// ---- per-slice start bins, built once per index: one workgroup per slice ---------------------------------------------------
// bins[j * (2 R + 2) + cl] = first slice-local row whose (start - min start) >> shift reaches cell cl (two cells per row);
// bins[.. + ncell] = rows of the slice.
__device__ __forceinline__ void cs_bins_body(unsigned char* cs_lds, uint32_t* wmin, int j, const int32_t* __restrict__ bound, const int32_t* __restrict__ b_start,
                                             const int2* __restrict__ ep, const int32_t* __restrict__ b_contig,
                                             const int32_t* __restrict__ seg, int R, unsigned short* __restrict__ bins,
                                             int4* __restrict__ smeta, int32_t* __restrict__ far_rows) {
    int32_t* l_start = reinterpret_cast<int32_t*>(cs_lds);                     // R
    unsigned short* l_bin = reinterpret_cast<unsigned short*>(l_start + R);    // 2 R + 2
    const int tid = threadIdx.x;
    const int r0 = bound[j];
    const int rk = bound[j + 1] - r0;
    if (rk <= 0) {                                                             // unused slot (uniform)
        if (tid == 0) { smeta[2 * j] = make_int4(0, 0, 0, INT32_MIN); smeta[2 * j + 1] = make_int4(0, -1, 0, r0); }
        return;
    }
    for (int i = tid; i < rk; i += CS_THREADS) l_start[i] = b_start[r0 + i];
    __syncthreads();
    const uint32_t s0 = flip(l_start[0]), s1 = flip(l_start[rk - 1]);
    const int ncell = 2 * rk;
    int bshift = 0;
    while ((unsigned long long)((s1 - s0) >> bshift) + 1ull > (unsigned long long)ncell) ++bshift;
    for (int i = tid; i <= ncell; i += CS_THREADS) l_bin[i] = i == ncell ? (unsigned short)rk : (unsigned short)0xffff;
    __syncthreads();
    for (int i = tid; i < rk; i += CS_THREADS) {
        const uint32_t c1 = (flip(l_start[i]) - s0) >> bshift;
        const bool head = i == 0 || ((flip(l_start[i - 1]) - s0) >> bshift) != c1;
        if (head) l_bin[c1] = (unsigned short)i;
    }
    __syncthreads();
    // empty cells take the next head: suffix minimum (heads ascend with the cell index)
    const int per_t = (ncell + 1 + CS_THREADS - 1) / CS_THREADS;
    const int c_lo = tid * per_t, c_hi = (c_lo + per_t) < (ncell + 1) ? (c_lo + per_t) : (ncell + 1);
    uint32_t mn = 0xffffffffu;
    for (int c = c_lo; c < c_hi; ++c) { const uint32_t x = l_bin[c]; mn = x < mn ? x : mn; }
    uint32_t run = sl_block_suffix_min_excl(mn, wmin);
    for (int c = c_hi - 1; c >= c_lo; --c) { const uint32_t x = l_bin[c]; run = x < run ? x : run; l_bin[c] = (unsigned short)run; }
    __syncthreads();
    unsigned short* out = bins + (size_t)j * (size_t)(2 * R + CS_BIN_STRIDE_PAD);
    for (int i = tid; i <= ncell; i += CS_THREADS) out[i] = l_bin[i];
    const int32_t cj = b_contig[r0];
    const int seg_a = seg[cj];
    if (tid == 0) {
        smeta[2 * j] = make_int4(l_start[0], bshift, ncell, r0 > seg_a ? ep[r0 - 1].y : INT32_MIN);
        smeta[2 * j + 1] = make_int4(seg_a, cj, rk, r0);
    }
    // Rows a short probe landing right behind them could NOT settle inside the branch-free window: the prefix max CS_WIN rows back
    // still reaches past the row's start.  Their share of the build side picks the join kernel (host: cs_far_share).
    int far = 0;
    for (int i = tid; i < rk; i += CS_THREADS) {
        const int p = r0 + i - CS_WIN;
        if (p >= seg_a && ep[p].y > l_start[i]) ++far;
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) far += __shfl_xor(far, d, kWave);
    // ONE atomic per workgroup (same-address atomics complete every ~12 ns whatever is in flight: a dense build side, where every
    // wavefront has something to add, paid 16 x 1000 of them = 0.1 - 0.2 ms for a 1 M-row index)
    __syncthreads();                                                           // wmin is free again
    if ((tid & (kWave - 1)) == 0) wmin[tid / kWave] = (uint32_t)far;
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < CS_WAVES; ++w) t += wmin[w];
        if (t) atomicAdd(far_rows, (int)t);
    }
}
__global__ __launch_bounds__(CS_THREADS) void k_cs_bins(const int32_t* __restrict__ bound, const int32_t* __restrict__ b_start,
                                                       const int2* __restrict__ ep, const int32_t* __restrict__ b_contig,
                                                       const int32_t* __restrict__ seg, int R, unsigned short* __restrict__ bins,
                                                       int4* __restrict__ smeta, int32_t* __restrict__ far_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
    __shared__ uint32_t wmin[CS_WAVES];
    cs_bins_body(cs_lds, wmin, (int)blockIdx.x, bound, b_start, ep, b_contig, seg, R, bins, smeta, far_rows);
}

// ---- partition, pass 1: per-chunk bucket histogram -----------------------------------------------------------------------------
struct CsTab {
    const unsigned long long* spl;
    const int4* cm;
    const uint32_t* cell;
};

__device__ __forceinline__ void cs_load_tab(const CsTab& tab, const CsGeom& g, unsigned long long* l_spl, int4* l_cm, uint32_t* l_cell, int threads) {
    for (int k = threadIdx.x; k < g.nb; k += threads) l_spl[k] = tab.spl[k];
    for (int k = threadIdx.x; k < g.ncells; k += threads) l_cell[k] = tab.cell[k];
    for (int k = threadIdx.x; k < g.n_contigs; k += threads) l_cm[k] = tab.cm[k];
}

template <bool STRICT>
__global__ __launch_bounds__(CS_THREADS) void k_cs_hist(CsTab tab, CsGeom g, const int32_t* __restrict__ pc, const int32_t* __restrict__ pe,
                                                       int64_t n, int chunk, int nchunks, bool vec_ok, uint32_t* __restrict__ blk_hist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
    // dynamic LDS: cm[CS_MAX_CONTIGS] | spl[nb] | cells[ncells] | hist[nb + 1]
    int4* l_cm = reinterpret_cast<int4*>(cs_lds);
    unsigned long long* l_spl = reinterpret_cast<unsigned long long*>(l_cm + CS_MAX_CONTIGS);
    uint32_t* l_cell = reinterpret_cast<uint32_t*>(l_spl + g.nb);
    uint32_t* h = l_cell + g.ncells;
    cs_load_tab(tab, g, l_spl, l_cm, l_cell, CS_THREADS);
    for (int k = threadIdx.x; k <= g.nb; k += CS_THREADS) h[k] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * chunk;
    const int64_t end = base + chunk < n ? base + chunk : n;
    for (int64_t i0 = base + (int64_t)threadIdx.x * 4; i0 < end; i0 += (int64_t)CS_THREADS * 4) {
        int32_t c[4], e[4];
        load_items_nt(pc, i0, end, vec_ok, -1, c);
        load_items_nt(pe, i0, end, vec_ok, 0, e);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i0 + k < end) atomicAdd(&h[cs_bucket<STRICT>(l_spl, l_cm, l_cell, g.nb, g.n_contigs, c[k], e[k])], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= g.nb; k += CS_THREADS) blk_hist[(int64_t)k * nchunks + blockIdx.x] = h[k];
}

// ---- partition, pass 2: unordered scatter of 12-byte records {start, end, row} --------------------------------------------------
// As k_slice_scatter_u: rank inside (tile, bucket) from one returning LDS atomic, records staged in LDS (three 4-byte
// planes: a 12-byte LDS store would need 16-byte alignment) at their bucket-sorted tile-local position, copied out as
// contiguous bucket runs with 12-byte stores.  A thread loads FOUR consecutive probes of each column with one 16-byte load.
struct CsPartLds { int cm, cell, spl, base, lstart, delta, cnt, rs, re, rr, d, wsum, total; };
__host__ __device__ inline CsPartLds cs_part_lds(int nb, int ncells, int n_contigs, int items) {
    const int tile = CS_THREADS * items;
    CsPartLds L;
    int o = 0;
    L.cm = o; o += 16 * ((n_contigs + 3) & ~3);
    L.spl = o; o += 8 * nb;
    L.cell = o; o += 4 * ncells;
    L.rs = (o + 15) & ~15; o = L.rs + 4 * tile;
    L.re = o; o += 4 * tile;
    L.rr = o; o += 4 * tile;
    L.base = o; o += 4 * (nb + 2);
    L.lstart = o; o += 4 * (nb + 2);
    L.delta = o; o += 4 * (nb + 2);
    L.cnt = o; o += 4 * (nb + 2);
    L.d = (o + 3) & ~3; o = L.d + 2 * tile;
    L.wsum = (o + 15) & ~15; o = L.wsum + 4 * 2 * CS_WAVES;
    L.total = o;
    return L;
}

// PITEMS probes per thread: tiles of 4096 (4) or 8192 (8) probes.  The record stores are what the scatter pays for (0.50 ms
// of 1.04 for config 3: ~25 M bucket runs of 48 bytes each); a tile twice as large makes every run twice as long.
// SAMPLED (round 4): no histogram pass in front.  The buckets own REGIONS of the record buffer sized from a 1 / 64 sample of the probe
// side (+ 25 % + a constant, k_cs_regions); a tile reserves the place of each of its bucket runs with one returning atomic on the
// bucket's cursor (blk_off = region starts, rcur = cursors), the join reads [region start, region start + cursor).  A run that does
// not fit its region raises bit 2 of the state word and is written over the region's start (in bounds: a region holds at least a
// tile): the host discards the call's result and redoes it with the exact, histogram-first partition.  Probes without a candidate
// row (bucket g.nb) are not stored at all.
template <bool STRICT, int PITEMS, bool SAMPLED, bool REC8 = false>
__global__ __launch_bounds__(CS_THREADS) void k_cs_scatter(CsTab tab, CsGeom g, const int32_t* __restrict__ pc, const int32_t* __restrict__ ps,
                                                          const int32_t* __restrict__ pe, const int32_t* __restrict__ row_id, int64_t n,
                                                          int chunk, int nchunks, bool vec_ok, const uint32_t* __restrict__ blk_off,
                                                          uint32_t* __restrict__ rcur, unsigned long long* __restrict__ state,
                                                          const int32_t* __restrict__ meta /* SAMPLED: [CS_META_FMT] = record format */,
                                                          int32_t* __restrict__ out /* 3 (or 2) int32 per record */, int ablate) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
    constexpr int TILE = CS_THREADS * PITEMS;
    const CsPartLds L = cs_part_lds(g.nb, g.ncells, g.n_contigs, PITEMS);
    unsigned long long* l_spl = reinterpret_cast<unsigned long long*>(cs_lds + L.spl);
    int4* l_cm = reinterpret_cast<int4*>(cs_lds + L.cm);
    uint32_t* l_cell = reinterpret_cast<uint32_t*>(cs_lds + L.cell);
    int32_t* l_rs = reinterpret_cast<int32_t*>(cs_lds + L.rs);
    int32_t* l_re = reinterpret_cast<int32_t*>(cs_lds + L.re);
    int32_t* l_rr = reinterpret_cast<int32_t*>(cs_lds + L.rr);
    uint32_t* base = reinterpret_cast<uint32_t*>(cs_lds + L.base);
    uint32_t* lstart = reinterpret_cast<uint32_t*>(cs_lds + L.lstart);
    uint32_t* delta = reinterpret_cast<uint32_t*>(cs_lds + L.delta);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(cs_lds + L.cnt);
    unsigned short* l_d = reinterpret_cast<unsigned short*>(cs_lds + L.d);
    int* wsum = reinterpret_cast<int*>(cs_lds + L.wsum);
    const int tid = threadIdx.x;
    const int nbk = g.nb + 1;
    // record format of this call (k_cs_regions, on the device): 0 = {start, end, row}; LB > 0 = {(end - slice minimum) << LB | (end - start), row}.
    // The host launches BOTH forms of the sampled scatter; the one the format does not name leaves at once (a runtime branch between the
    // two forms inside one kernel keeps the ends alive next to the packed words: 6 spilled VGPRs at this kernel's 128-register ceiling).
    static_assert(!REC8 || SAMPLED, "8-byte records belong to the sampled partition");
    const int lb = SAMPLED ? meta[CS_META_FMT] : 0;                             // uniform
    if (SAMPLED && (lb != 0) != REC8) return;
    cs_load_tab(tab, g, l_spl, l_cm, l_cell, CS_THREADS);
    // exact: base[b] = running global offset of bucket b for this chunk; SAMPLED: base[b] = start of bucket b's region (b <= g.nb)
    for (int k = tid; k < nbk + 1; k += CS_THREADS) { base[k] = SAMPLED ? blk_off[k < nbk ? k : nbk - 1] : (k < nbk ? blk_off[(int64_t)k * nchunks + blockIdx.x] : 0u); cnt[k] = 0; }
    __syncthreads();
    const int64_t cbase = (int64_t)blockIdx.x * chunk;
    const int64_t cend = cbase + chunk < n ? cbase + chunk : n;
    // thread t holds the probes t * 4 .. t * 4 + 3 of every 4096-probe half of the tile (16-byte column loads)
    int32_t nc[PITEMS], ns[PITEMS], ne[PITEMS], nr[PITEMS];
    auto load_tile = [&](int64_t tbase) {
#pragma unroll
        for (int h = 0; h < PITEMS / 4; ++h) {
            const int64_t i0 = tbase + (int64_t)h * (CS_THREADS * 4) + (int64_t)tid * 4;
            load_items_nt(pc, i0, cend, vec_ok, -1, *reinterpret_cast<int32_t(*)[4]>(nc + 4 * h));
            load_items_nt(ps, i0, cend, vec_ok, 0, *reinterpret_cast<int32_t(*)[4]>(ns + 4 * h));
            load_items_nt(pe, i0, cend, vec_ok, 0, *reinterpret_cast<int32_t(*)[4]>(ne + 4 * h));
            if (row_id) load_items_nt(row_id, i0, cend, vec_ok, -1, *reinterpret_cast<int32_t(*)[4]>(nr + 4 * h));
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) nr[4 * h + j] = (int32_t)(i0 + j);
            }
        }
    };
    load_tile(cbase);
    int tix = 0;
    for (int64_t tbase = cbase; tbase < cend; tbase += TILE, ++tix) {
        int32_t c[PITEMS], s[PITEMS], e[PITEMS], r[PITEMS];
#pragma unroll
        for (int j = 0; j < PITEMS; ++j) { c[j] = nc[j]; s[j] = ns[j]; e[j] = ne[j]; r[j] = nr[j]; }
        if (tbase + TILE < cend) load_tile(tbase + TILE);                      // next tile's columns in flight during this one
        const int tile_n = (int)((cend - tbase) < (int64_t)TILE ? (cend - tbase) : (int64_t)TILE);
        uint32_t d[PITEMS], rank[PITEMS];
#pragma unroll
        for (int j = 0; j < PITEMS; ++j) {
            const bool valid = (j / 4) * (CS_THREADS * 4) + tid * 4 + (j & 3) < tile_n;
            // (profiling only, IVJ_SLICE_ABLATE: 256 no record stores, 1024 hashed bucket ids instead of the table lookup; results are wrong)
            if (ablate & 1024) d[j] = !valid ? 0u : (uint32_t)(((uint32_t)e[j] * 2654435761u) >> 12) % (uint32_t)g.nb;
            else d[j] = !valid ? 0u : cs_bucket<STRICT>(l_spl, l_cm, l_cell, g.nb, g.n_contigs, c[j], e[j]);
            rank[j] = valid ? atomicAdd(&cnt[d[j]], 1u) : 0u;
            if constexpr (REC8) {                                               // the packed word takes the place of the start
                uint32_t w0 = 0u;
                if (valid && d[j] < (uint32_t)g.nb) {
                    const uint32_t off = (uint32_t)e[j] - (uint32_t)unflip((uint32_t)l_spl[d[j]]);      // >= 0: the slice's first row starts below the end
                    const uint32_t len = (uint32_t)e[j] - (uint32_t)s[j];
                    if (e[j] < s[j] || (len >> lb) != 0u || (off >> (32 - lb)) != 0u) atomicOr(state + 1, CS_STATE_REC8);   // (redo with 12-byte records)
                    w0 = (off << lb) | len;
                }
                s[j] = (int32_t)w0;
            }
        }
        __syncthreads();                                                        // (A) bucket counts of the tile complete
        // thread t owns OWN consecutive buckets: tile-local starts, copy-out deltas, running global offsets; counters cleared
        constexpr int OWN = (SL_MAX_BUCKETS + 1 + CS_THREADS - 1) / CS_THREADS;
        int x[OWN];
        int xs = 0;
#pragma unroll
        for (int q = 0; q < OWN; ++q) {
            const int b = OWN * tid + q;
            x[q] = 0;
            if (b < nbk) { x[q] = (int)cnt[b]; cnt[b] = 0; }
            xs += x[q];
        }
        long long tsum;
        int pre = (int)sl_block_exclusive_sum_i32<CS_WAVES>(xs, wsum + (tix & 1) * CS_WAVES, &tsum);      // (B)
        // SAMPLED, split-phase (round 5): the returning atomic that reserves a run's place in its bucket's region is only ISSUED here;
        // its answer is needed for the copy-out deltas alone, so the placement into LDS runs while it is in flight (the round-4 form
        // waited for it -- twice in a row for a thread's two buckets, ~ 3 us of an 18-us tile -- before barrier (C))
        uint32_t got[OWN];
#pragma unroll
        for (int q = 0; q < OWN; ++q) {
            const int b = OWN * tid + q;
            got[q] = 0u;
            if (b < nbk) {
                lstart[b] = (uint32_t)pre;
                if constexpr (SAMPLED) {
                    if (x[q] > 0 && b < g.nb) got[q] = atomicAdd(&rcur[(size_t)b * CS_CUR_STRIDE], (uint32_t)x[q]);
                } else { delta[b] = base[b] - (uint32_t)pre; base[b] += (uint32_t)x[q]; }
            }
            pre += x[q];
        }
        __syncthreads();                                                        // (C)
#pragma unroll
        for (int j = 0; j < PITEMS; ++j) {
            if ((j / 4) * (CS_THREADS * 4) + tid * 4 + (j & 3) < tile_n) {
                const uint32_t pos = lstart[d[j]] + rank[j];
                l_rs[pos] = s[j]; l_rr[pos] = r[j];
                if constexpr (!REC8) l_re[pos] = e[j];
                l_d[pos] = (unsigned short)d[j];
            }
        }
        if constexpr (SAMPLED) {
#pragma unroll
            for (int q = OWN - 1; q >= 0; --q) {
                const int b = OWN * tid + q;
                pre -= x[q];
                if (b < nbk) {
                    uint32_t at = base[b];                                      // region start (an overflowing run lands here, in bounds)
                    if (x[q] > 0 && b < g.nb) {
                        const uint32_t cap = base[b + 1] - at;
                        if (got[q] + (uint32_t)x[q] <= cap) at += got[q];
                        else atomicOr(state + 1, 4ull);
                    }
                    delta[b] = at - (uint32_t)pre;
                }
            }
        }
        __syncthreads();                                                        // (D) tile sorted in LDS
#pragma unroll
        for (int j = 0; j < PITEMS; ++j) {
            const int il = j * CS_THREADS + tid;
            if (il < tile_n && !(ablate & 256) && !(SAMPLED && l_d[il] == (unsigned short)g.nb)) {
                uint32_t oi = (uint32_t)il + delta[l_d[il]];
                if (ablate & 2048) oi &= (1u << 22) - 1u;                      // profiling only: every store lands in the first 48 MB (address-translation probe)
                if constexpr (REC8) {
                    cs_rec8 v; v.x = l_rs[il]; v.y = l_rr[il];
                    *reinterpret_cast<cs_rec8*>(out + 2 * (int64_t)oi) = v;
                } else {
                    cs_rec v; v.x = l_rs[il]; v.y = l_re[il]; v.z = l_rr[il];
                    *reinterpret_cast<cs_rec*>(out + 3 * (int64_t)oi) = v;
                }
            }
        }
        // no barrier here: the next tile's barrier (A) separates this copy-out from the next placement
    }
}

// ---- the sampled scatter of 8-byte records on 12 288-probe tiles (round 6) --------------------------------------------------------------
// What the scatter pays for is the SHAPE of its record stores (tools/micro/write_calib.hip, profiles/r06/write_calibration.txt: runs of
// 8 records at a random 8-byte offset cost 1.36 x their bytes in 64 / 32-byte write requests, runs of 12 records 1.25 x, of 16 1.18 x) and
// its per-tile fixed work (four barriers, a scan over the ~ 1000 buckets): 4096-probe tiles 1.12 ms, 8192-probe tiles 0.70 ms for config 3
// (profiles/r06/ab_scatter_tile_4096.json).  This form takes 12 288 probes per tile -- runs of ~ 12 records / 93 bytes at 1040 buckets:
//   * LDS: two 4-byte staging planes (packed word, row) + the 2-byte bucket plane = 10 bytes per probe (the 8192-probe form carries a third
//     plane for the 12-byte records' ends) and no copy of the region starts (read from global memory, once per bucket and tile);
//   * registers: the tile's columns are consumed out of the registers they were loaded into, {bucket, rank} share one word, the packed
//     word is built while the bucket is looked up, the row is recomputed at placement time (or loaded there, when the side brings row ids),
//     and the NEXT tile's columns are requested only once this tile's are dead -- 12 probes per thread inside the 128-register budget.
// Same protocol as k_cs_scatter<.., SAMPLED = true, REC8 = true>: the host queues it INSTEAD of that kernel where cs_part12_lds fits the
// LDS (the plan's part_items = 12), next to the 12-byte form that runs when the device-side format word says so.
// items = 16 (16 384-probe tiles, sides WITHOUT row ids): the row plane holds the 2-byte tile-local element index (row = tile base + index)
// and there is no bucket plane -- the copy-out walks the bucket runs instead of the elements: 6 bytes of staging per probe.
struct CsPart12Lds { int cm, cell, spl, lstart, delta, cnt, rs, rr, d, wsum, total; };
__host__ __device__ inline CsPart12Lds cs_part12_lds(int nb, int ncells, int n_contigs, int items = 12) {
    const int tile = CS_THREADS * items;
    CsPart12Lds L;
    int o = 0;
    L.cm = o; o += 16 * ((n_contigs + 3) & ~3);
    L.spl = o; o += 8 * nb;
    L.cell = o; o += 4 * ncells;
    L.rs = (o + 15) & ~15; o = L.rs + 4 * tile;
    L.rr = o; o += (items == 16 ? 2 : 4) * tile;
    L.lstart = o; o += 4 * (nb + 4);
    L.delta = o; o += 4 * (nb + 4);
    L.cnt = o; o += 4 * (nb + 4);
    L.d = (o + 3) & ~3; o = L.d + (items == 16 ? 0 : 2 * tile);
    L.wsum = (o + 15) & ~15; o = L.wsum + 4 * 2 * CS_WAVES;
    L.total = o;
    return L;
}

template <bool STRICT, int PITEMS>
__global__ __launch_bounds__(CS_THREADS) void k_cs_scatter12k(CsTab tab, CsGeom g, const int32_t* __restrict__ pc, const int32_t* __restrict__ ps,
                                                             const int32_t* __restrict__ pe, const int32_t* __restrict__ row_id, int64_t n,
                                                             int chunk, int nchunks, bool vec_ok, const uint32_t* __restrict__ rstart,
                                                             uint32_t* __restrict__ rcur, unsigned long long* __restrict__ state,
                                                             const int32_t* __restrict__ meta, int32_t* __restrict__ out, int ablate,
                                                             unsigned long long* __restrict__ ptrace /* IVJ_CS_PTRACE (diagnosis): phase stamps of every workgroup's second tile */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
    constexpr int TILE = CS_THREADS * PITEMS;
    constexpr bool RUNS = PITEMS == 16;                                         // 16 384-probe tiles: 2-byte row plane, copy-out by bucket runs (no row ids)
    static_assert(PITEMS == 12 || PITEMS == 16, "tile sizes of this kernel");
    const CsPart12Lds L = cs_part12_lds(g.nb, g.ncells, g.n_contigs, PITEMS);
    unsigned long long* l_spl = reinterpret_cast<unsigned long long*>(cs_lds + L.spl);
    int4* l_cm = reinterpret_cast<int4*>(cs_lds + L.cm);
    uint32_t* l_cell = reinterpret_cast<uint32_t*>(cs_lds + L.cell);
    int32_t* l_rs = reinterpret_cast<int32_t*>(cs_lds + L.rs);
    int32_t* l_rr = reinterpret_cast<int32_t*>(cs_lds + L.rr);
    unsigned short* l_ri = reinterpret_cast<unsigned short*>(cs_lds + L.rr);    // RUNS: tile-local element index instead of the row
    uint32_t* lstart = reinterpret_cast<uint32_t*>(cs_lds + L.lstart);
    uint32_t* delta = reinterpret_cast<uint32_t*>(cs_lds + L.delta);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(cs_lds + L.cnt);
    unsigned short* l_d = reinterpret_cast<unsigned short*>(cs_lds + L.d);
    int* wsum = reinterpret_cast<int*>(cs_lds + L.wsum);
    const int tid = threadIdx.x;
    const int nbk = g.nb + 1;
    const int lb = meta[CS_META_FMT];                                           // uniform: 0 = this call's records are the 12-byte ones (the other launch)
    if (lb == 0) return;
    cs_load_tab(tab, g, l_spl, l_cm, l_cell, CS_THREADS);
    for (int k = tid; k < nbk + 1; k += CS_THREADS) cnt[k] = 0;
    if (RUNS && row_id) return;                                                  // (host contract: sides with row ids take the 12 288-probe form)
    __syncthreads();
    const int64_t cbase = (int64_t)blockIdx.x * chunk;
    const int64_t cend = cbase + chunk < n ? cbase + chunk : n;
    // thread t holds the probes t * 4 .. t * 4 + 3 of every 4096-probe third of the tile (16-byte column loads).  The addresses are a
    // UNIFORM tile pointer + a 32-bit lane offset (scalar base + vector offset in the instruction): nine 64-bit per-lane addresses kept
    // across the tile loop were what spilled in the first version of this kernel
    int32_t nc[PITEMS], ns[PITEMS], ne[PITEMS];
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    auto load4 = [&](const int32_t* __restrict__ col /* uniform */, uint32_t e0, uint32_t rem, int32_t fill, int32_t* dst) {
        if (vec_ok && e0 + 4u <= rem) {
            const v4i_t v = __builtin_nontemporal_load(reinterpret_cast<const v4i_t*>(col + e0));
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) dst[u] = (e0 + (uint32_t)u < rem) ? __builtin_nontemporal_load(col + e0 + u) : fill;
        }
    };
    auto load_tile = [&](int64_t tbase) {
        const uint32_t rem = (uint32_t)((cend - tbase) < (int64_t)TILE ? (cend - tbase) : (int64_t)TILE);
        const int32_t *bc = pc + tbase, *bs = ps + tbase, *be = pe + tbase;
#pragma unroll
        for (int h = 0; h < PITEMS / 4; ++h) {
            const uint32_t e0 = (uint32_t)(h * (CS_THREADS * 4) + tid * 4);
            load4(bc, e0, rem, -1, nc + 4 * h);
            load4(bs, e0, rem, 0, ns + 4 * h);
            load4(be, e0, rem, 0, ne + 4 * h);
        }
    };
    load_tile(cbase);
    int tix = 0;
    for (int64_t tbase = cbase; tbase < cend; tbase += TILE, ++tix) {
        const int tile_n = (int)((cend - tbase) < (int64_t)TILE ? (cend - tbase) : (int64_t)TILE);
        if (ptrace && (tix == 1 || tix == 2) && tid == 0) ptrace[8 * (size_t)blockIdx.x + (tix == 1 ? 0 : 5)] = wall_clock64();
        // packed record word; bucket | rank << 11.  RUNS (16 probes per lane): the word is built at PLACEMENT time from the columns, which
        // stay in their registers until then, and the next tile's columns are requested after the placement (their flight overlaps the
        // copy-out and the tile's last barrier) -- sixteen packed words next to sixteen prefetched probes do not fit 128 registers
        uint32_t w0[RUNS ? 1 : PITEMS], dr[PITEMS];
        bool bad = false;                                                       // a probe of this lane does not fit the 8-byte form
        static_assert(SL_MAX_BUCKETS + 1 <= (1 << 11) && TILE <= (1 << 21), "bucket and rank share one word");
#pragma unroll
        for (int j = 0; j < PITEMS; ++j) {
            // (no test for the tile's end: a lane beyond it holds the fill values -- contig -1 -- and counts into bucket nb, which is never copied)
            const uint32_t d = cs_bucket_flat<STRICT>(l_spl, l_cm, l_cell, g.nb, g.n_contigs, nc[j], ne[j]);
            const uint32_t rank = atomicAdd(&cnt[d], 1u);
            if constexpr (!RUNS) {
                const uint32_t off = (uint32_t)ne[j] - (uint32_t)unflip((uint32_t)l_spl[d < (uint32_t)g.nb ? d : 0u]);   // >= 0: the slice's first row starts below the end
                const uint32_t len = (uint32_t)ne[j] - (uint32_t)ns[j];
                bad = bad || (d < (uint32_t)g.nb && (ne[j] < ns[j] || (len >> lb) != 0u || (off >> (32 - lb)) != 0u));
                w0[j] = (off << lb) | len;                                     // (bucket nb: garbage, never copied)
            }
            dr[j] = d | (rank << 11);
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);               // (four lookups in flight are enough: twelve interleaved ones cost 14 spilled registers)
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!RUNS) {
            if (tbase + TILE < cend) load_tile(tbase + TILE);                  // this tile's columns are dead: the next tile's travel during the rest of this one
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                                        // (A) bucket counts of the tile complete
        if (ptrace && tix == 1 && tid == 0) ptrace[8 * (size_t)blockIdx.x + 1] = wall_clock64();
        // (the lane's index is made opaque once per tile: per-lane addresses derived from it -- the region cursors' 64-bit ones, the 3 x 12
        // LDS addresses of the copy-out -- are otherwise loop invariants the compiler keeps in registers across the tile loop, and spills)
        int tv = tid;
        asm volatile("" : "+v"(tv));
        constexpr int OWN = (SL_MAX_BUCKETS + 1 + CS_THREADS - 1) / CS_THREADS;
        int x[OWN];
        int xs = 0;
#pragma unroll
        for (int q = 0; q < OWN; ++q) {
            const int b = OWN * tv + q;
            x[q] = 0;
            if (b < nbk) { x[q] = (int)cnt[b]; cnt[b] = 0; }
            xs += x[q];
        }
        // (B) workgroup exclusive sum with the DPP wavefront scan (no ds_bpermute lane-address registers to keep across the tile loop)
        int pre;
        {
            int* part = wsum + (tix & 1) * CS_WAVES;
            const int inc = wave_incl_sum_dpp(xs);
            if ((tv & (kWave - 1)) == kWave - 1) part[tv / kWave] = inc;
            __syncthreads();
            int below = 0;
#pragma unroll
            for (int k = 0; k < CS_WAVES; k += 4) {
                const int4 pw = *reinterpret_cast<const int4*>(part + k);
                const int w = tv / kWave;
                below += (k < w ? pw.x : 0) + (k + 1 < w ? pw.y : 0) + (k + 2 < w ? pw.z : 0) + (k + 3 < w ? pw.w : 0);
            }
            pre = below + (inc - xs);
        }
        uint32_t got[OWN], r_at[OWN], r_cap[OWN];
#pragma unroll
        for (int q = 0; q < OWN; ++q) {
            const int b = OWN * tv + q;
            got[q] = 0u; r_at[q] = 0u; r_cap[q] = 0u;
            if (b < nbk) {
                lstart[b] = (uint32_t)pre;
                if (RUNS && b == nbk - 1) lstart[nbk] = (uint32_t)(pre + x[q]);  // (the runs' copy-out reads bucket b's end at lstart[b + 1])
                // split-phase: the cursor's answer and the region's bounds (global memory: this form keeps no LDS copy of them) are
                // requested here and used after the placement
                r_at[q] = rstart[b < g.nb ? b : g.nb];
                if (x[q] > 0 && b < g.nb) { r_cap[q] = rstart[b + 1]; got[q] = atomicAdd(&rcur[(size_t)b * CS_CUR_STRIDE], (uint32_t)x[q]); }
            }
            pre += x[q];
        }
        __syncthreads();                                                        // (C)
        if (ptrace && tix == 1 && tid == 0) ptrace[8 * (size_t)blockIdx.x + 2] = wall_clock64();
#pragma unroll
        for (int h = 0; h < PITEMS / 4; ++h) {
            const int e0 = h * (CS_THREADS * 4) + tv * 4;
            int32_t rw[4];
            if constexpr (!RUNS) {
                if (row_id) load4(row_id + tbase, (uint32_t)e0, (uint32_t)tile_n, -1, rw);
                else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) rw[u] = (int32_t)(tbase + e0 + u);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = 4 * h + u;
                // (every lane places: the lanes beyond the tile's end were ranked into bucket nb)
                const uint32_t d = dr[j] & 2047u;
                const uint32_t pos = lstart[d] + (dr[j] >> 11);
                if constexpr (RUNS) {
                    const uint32_t off = (uint32_t)ne[j] - (uint32_t)unflip((uint32_t)l_spl[d < (uint32_t)g.nb ? d : 0u]);
                    const uint32_t len = (uint32_t)ne[j] - (uint32_t)ns[j];
                    bad = bad || (d < (uint32_t)g.nb && (ne[j] < ns[j] || (len >> lb) != 0u || (off >> (32 - lb)) != 0u));
                    l_rs[pos] = (int32_t)((off << lb) | len); l_ri[pos] = (unsigned short)(e0 + u);
                } else { l_rs[pos] = (int32_t)w0[j]; l_rr[pos] = rw[u]; l_d[pos] = (unsigned short)d; }
            }
        }
        if (bad) atomicOr(state + 1, CS_STATE_REC8);                            // (redo with 12-byte records)
        if (ptrace && tix == 1 && tid == 0) ptrace[8 * (size_t)blockIdx.x + 6] = wall_clock64();
        if constexpr (RUNS) {
            __builtin_amdgcn_sched_barrier(0);
            if (tbase + TILE < cend) load_tile(tbase + TILE);                  // (the columns were consumed by the placement)
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = OWN - 1; q >= 0; --q) {
            const int b = OWN * tv + q;
            pre -= x[q];
            if (b < nbk) {
                uint32_t at = r_at[q];                                          // region start (an overflowing run lands here, in bounds)
                if (x[q] > 0 && b < g.nb) {
                    const uint32_t cap = r_cap[q] - at;
                    if (got[q] + (uint32_t)x[q] <= cap) at += got[q];
                    else atomicOr(state + 1, 4ull);
                }
                delta[b] = at - (uint32_t)pre;
            }
        }
        if (ptrace && tix == 1 && tid == 0) ptrace[8 * (size_t)blockIdx.x + 7] = wall_clock64();
        __syncthreads();                                                        // (D) tile sorted in LDS
        if (ptrace && tix == 1 && tid == 0) ptrace[8 * (size_t)blockIdx.x + 3] = wall_clock64();
        if constexpr (RUNS) {
            // copy-out by RUNS: sixteen lanes per bucket run, four runs per wavefront step (a run holds ~ 16 records at 1040 buckets);
            // bucket g.nb (no candidate row) is never copied
            if (!(ablate & 256)) {
                if (ablate & 32768) {                                           // (IVJ_SLICE_ABLATE bit 32768, A/B: one 8-byte record per lane and store: 0.554 - 0.571 against 0.547 - 0.548 ms)
                const int grp = (tv & (kWave - 1)) >> 4, sub = tv & 15;
                for (int b0 = (tv / kWave) * 4; b0 < g.nb; b0 += CS_WAVES * 4) {
                    const int b = b0 + grp;
                    uint32_t rs0 = 0, re0 = 0, dl = 0;
                    if (b < g.nb) { rs0 = lstart[b]; re0 = lstart[b + 1]; dl = delta[b]; }
                    for (uint32_t kk = rs0 + (uint32_t)sub; kk < re0; kk += 16u) {
                        cs_rec8 v; v.x = l_rs[kk]; v.y = (int32_t)(tbase + (int64_t)l_ri[kk]);
                        *reinterpret_cast<cs_rec8*>(out + 2 * (int64_t)(kk + dl)) = v;
                    }
                }
                } else {
                // eight lanes per run, TWO records = 16 bytes per lane and store, pairs aligned to 16 bytes in the record buffer
                const int grp = (tv & (kWave - 1)) >> 3, sub = tv & 7;
                typedef int cs_v4 __attribute__((ext_vector_type(4)));
                for (int b0 = (tv / kWave) * 8; b0 < g.nb; b0 += CS_WAVES * 8) {
                    const int b = b0 + grp;
                    uint32_t rs0 = 0, re0 = 0, dl = 0;
                    if (b < g.nb) { rs0 = lstart[b]; re0 = lstart[b + 1]; dl = delta[b]; }
                    // (signed: the pair in front of an odd first record starts one record before the run)
                    for (int kk = (int)rs0 - (int)((rs0 + dl) & 1u) + 2 * sub; kk < (int)re0; kk += 16) {
                        const bool v0 = kk >= (int)rs0, v1 = kk + 1 < (int)re0;
                        const int k0 = v0 ? kk : kk + 1, k1 = v1 ? kk + 1 : kk;
                        const int32_t w0 = l_rs[k0], r0 = (int32_t)(tbase + (int64_t)l_ri[k0]);
                        const int32_t w1 = l_rs[k1], r1 = (int32_t)(tbase + (int64_t)l_ri[k1]);
                        int32_t* dst = out + 2 * ((int64_t)kk + (int64_t)dl);
                        if (v0 && v1) { cs_v4 v = {w0, r0, w1, r1}; *reinterpret_cast<cs_v4*>(dst) = v; }
                        else if (v0) { cs_rec8 v; v.x = w0; v.y = r0; *reinterpret_cast<cs_rec8*>(dst) = v; }
                        else if (v1) { cs_rec8 v; v.x = w1; v.y = r1; *reinterpret_cast<cs_rec8*>(dst + 2) = v; }
                    }
                }
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < PITEMS; ++j) {
            const int il = j * CS_THREADS + tv;
            if (il < tile_n && !(ablate & 256) && l_d[il] != (unsigned short)g.nb) {
                const uint32_t oi = (uint32_t)il + delta[l_d[il]];
                cs_rec8 v; v.x = l_rs[il]; v.y = l_rr[il];
                *reinterpret_cast<cs_rec8*>(out + 2 * (int64_t)oi) = v;
            }
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);               // (four records in flight per lane: all twelve at once spill next to the prefetched columns)
        }
        }
        // no barrier here: the next tile's barrier (A) separates this copy-out from the next placement
        if (ptrace && tix == 1 && tid == 0) ptrace[8 * (size_t)blockIdx.x + 4] = wall_clock64();
    }
}

// ---- SAMPLED partition: region sizes from 1 / 64 of the probe side ---------------------------------------------------------------
// Sample = groups of CS_SGROUP consecutive probes (one 32-byte sector per column) every CS_SGROUP * CS_SRATE probes; gh[b] += sampled
// probes of bucket b.  A few hundred workgroups, the bucket table in LDS as in k_cs_hist.
constexpr int CS_SGROUP = 8, CS_SRATE = 64;
// gh[nb + 1] / gh[nb + 2] (round 5): the sample's largest probe length (end - start; 2^31 - 1 for an inverted row) and its largest
// distance of a probe's end from its slice's smallest start -- what decides whether the call's records take the 8-byte form.
template <bool STRICT>
__device__ __forceinline__ void cs_sample_body(unsigned char* cs_lds, unsigned bid, unsigned nblocks, CsTab tab, CsGeom g, const int32_t* __restrict__ pc,
                                               const int32_t* __restrict__ ps, const int32_t* __restrict__ pe, int64_t n, uint32_t* __restrict__ gh) {
    int4* l_cm = reinterpret_cast<int4*>(cs_lds);
    unsigned long long* l_spl = reinterpret_cast<unsigned long long*>(l_cm + CS_MAX_CONTIGS);
    uint32_t* l_cell = reinterpret_cast<uint32_t*>(l_spl + g.nb);
    uint32_t* h = l_cell + g.ncells;                                            // nb + 1 counts, then the two maxima
    cs_load_tab(tab, g, l_spl, l_cm, l_cell, CS_THREADS);
    for (int k = threadIdx.x; k <= g.nb + 2; k += CS_THREADS) h[k] = 0;
    __syncthreads();
    uint32_t mlen = 0, moff = 0;
    const int64_t n_groups = (n + (int64_t)CS_SGROUP * CS_SRATE - 1) / ((int64_t)CS_SGROUP * CS_SRATE);
    for (int64_t t = (int64_t)bid * CS_THREADS + threadIdx.x; t < n_groups * CS_SGROUP; t += (int64_t)nblocks * CS_THREADS) {
        const int64_t i = (t / CS_SGROUP) * ((int64_t)CS_SGROUP * CS_SRATE) + (t % CS_SGROUP);
        if (i < n) {
            const int32_t qe = pe[i], qs = ps[i];
            const uint32_t b = cs_bucket<STRICT>(l_spl, l_cm, l_cell, g.nb, g.n_contigs, pc[i], qe);
            atomicAdd(&h[b], 1u);
            if (b < (uint32_t)g.nb) {
                const uint32_t len = qe >= qs ? (uint32_t)qe - (uint32_t)qs : 0x7fffffffu;
                const uint32_t off = (uint32_t)qe - (uint32_t)unflip((uint32_t)l_spl[b]);       // >= 0 by the bucket's definition
                mlen = len > mlen ? len : mlen;
                moff = off > moff ? off : moff;
            }
        }
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        const uint32_t a = __shfl_xor(mlen, d, kWave), b = __shfl_xor(moff, d, kWave);
        mlen = a > mlen ? a : mlen; moff = b > moff ? b : moff;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) { atomicMax(&h[g.nb + 1], mlen); atomicMax(&h[g.nb + 2], moff); }
    __syncthreads();
    for (int k = threadIdx.x; k <= g.nb; k += CS_THREADS) if (h[k]) atomicAdd(&gh[k], h[k]);
    if (threadIdx.x == 0) { atomicMax(&gh[g.nb + 1], h[g.nb + 1]); atomicMax(&gh[g.nb + 2], h[g.nb + 2]); }
}
template <bool STRICT>
__global__ __launch_bounds__(CS_THREADS) void k_cs_sample_hist(CsTab tab, CsGeom g, const int32_t* __restrict__ pc, const int32_t* __restrict__ ps,
                                                              const int32_t* __restrict__ pe, int64_t n, uint32_t* __restrict__ gh) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
    cs_sample_body<STRICT>(cs_lds, blockIdx.x, gridDim.x, tab, g, pc, ps, pe, n, gh);
}
// Round 5: the first call on a fresh index runs the per-slice bins (once per index) and the probe sample (once per call) as ONE launch --
// both only need k_cs_prep's tables and do not depend on each other: workgroups [0, n_bins) build bins, the rest sample.
template <bool STRICT>
__global__ __launch_bounds__(CS_THREADS) void k_cs_bins_sample(int n_bins, const int32_t* __restrict__ bound, const int32_t* __restrict__ b_start,
                                                              const int2* __restrict__ ep, const int32_t* __restrict__ b_contig,
                                                              const int32_t* __restrict__ seg, int R, unsigned short* __restrict__ bins,
                                                              int4* __restrict__ smeta, int32_t* __restrict__ far_rows,
                                                              CsTab tab, CsGeom g, const int32_t* __restrict__ pc, const int32_t* __restrict__ ps,
                                                              const int32_t* __restrict__ pe, int64_t n, uint32_t* __restrict__ gh) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
    __shared__ uint32_t wmin[CS_WAVES];
    if ((int)blockIdx.x < n_bins) cs_bins_body(cs_lds, wmin, (int)blockIdx.x, bound, b_start, ep, b_contig, seg, R, bins, smeta, far_rows);
    else cs_sample_body<STRICT>(cs_lds, blockIdx.x - (unsigned)n_bins, gridDim.x - (unsigned)n_bins, tab, g, pc, ps, pe, n, gh);
}

// Regions of the record buffer: cap_b = 1.25 x (sampled count x CS_SRATE) + slack, starts aligned to 32 records (three 128-byte lines);
// rstart[nb] = records reserved in total (bucket nb, "no candidate row", owns no region).  slack >= one partition tile, so that a run
// which does not fit can be parked at its region's start without leaving it.  Clears the cursors.  One workgroup.
// Record format of the call (round 5), meta[CS_META_FMT]: 0 = 12-byte records {start, end, row}; LB > 0 = 8-byte records
// {(end - smallest start of the probe's slice) << LB | (end - start), row}.  A bucket's probes all END inside one slice's start span
// (that is what the bucket is), and probes are short: for the benchmark shapes 22 + 8 bits.  The choice is made HERE, on the device,
// from the sample's maxima with margins (length x 2 + 64, offset x 1.25 + 4096); a probe that does not fit after all (an outlier the
// sample missed) raises bit 8 of the state word in the scatter and the host redoes the call with 12-byte records: exactness never
// rests on the sample.  allow8 = 0: the caller wants 12-byte records (the redo, IVJ_CS_REC8=0).
__global__ __launch_bounds__(CS_THREADS) void k_cs_regions(const uint32_t* __restrict__ gh, int nb, uint32_t slack, int allow8, uint32_t* __restrict__ rstart,
                                                          uint32_t* __restrict__ rcur, int32_t* __restrict__ meta, const int32_t* __restrict__ far_src,
                                                          uint32_t* __restrict__ hw, uint32_t hw_seq) {
    if (threadIdx.x == 0 && hw) {
        // host words [4], [5]: the far-row count the bins kernel (queued in front of this one) left in the index, for cs_resolve_tables
        hw_store(hw + 4, (uint32_t)__hip_atomic_load(far_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        hw_store(hw + 5, hw_seq);
    }
    if (threadIdx.x == 0) {
        int lb = 0;
        if (allow8) {
            const unsigned long long ml = (unsigned long long)gh[nb + 1] * 2ull + 64ull;
            const unsigned long long mo = (unsigned long long)gh[nb + 2] + gh[nb + 2] / 4 + 4096ull;
            const int lbits = ml > 0xffffffffull ? 33 : cs_bits_for((uint32_t)ml);
            const int obits = mo > 0xffffffffull ? 33 : cs_bits_for((uint32_t)mo);
            if (lbits + obits <= 32) lb = 32 - obits;                           // the spare bits go to the length
        }
        meta[CS_META_FMT] = lb;
    }
    __shared__ __attribute__((aligned(16))) int wsum[CS_WAVES];
    constexpr int OWN = (SL_MAX_BUCKETS + 1 + CS_THREADS - 1) / CS_THREADS;
    uint32_t cap[OWN];
    long long mine = 0;
#pragma unroll
    for (int q = 0; q < OWN; ++q) {
        const int b = OWN * threadIdx.x + q;
        cap[q] = 0;
        if (b < nb) {
            const unsigned long long est = (unsigned long long)gh[b] * CS_SRATE;
            unsigned long long c = est + est / 4 + slack;
            c = (c + 31ull) & ~31ull;
            cap[q] = (uint32_t)c;
        }
        mine += cap[q];
    }
    long long total;
    long long pre = sl_block_exclusive_sum_i32<CS_WAVES>((int)mine, wsum, &total);      // (capacities sum below 2^31 records: checked on the host)
#pragma unroll
    for (int q = 0; q < OWN; ++q) {
        const int b = OWN * threadIdx.x + q;
        if (b < nb) { rstart[b] = (uint32_t)pre; rcur[(size_t)b * CS_CUR_STRIDE] = 0u; }
        pre += cap[q];
    }
    if (threadIdx.x == 0) { rstart[nb] = (uint32_t)total; rcur[(size_t)nb * CS_CUR_STRIDE] = 0u; }
}

// chunk table of the join from the regions and their final cursors (k_slice_chunks' counterpart): bstart / bend per bucket, the
// workgroup -> (bucket, chunk) map, meta[0] = number of join workgroups.  One workgroup.
__global__ __launch_bounds__(SL_THREADS) void k_cs_chunks_sampled(const uint32_t* __restrict__ rstart, const uint32_t* __restrict__ rcur, int nb,
                                                                 int jchunk, uint32_t* __restrict__ bstart, uint32_t* __restrict__ bend,
                                                                 int32_t* __restrict__ meta, int2* __restrict__ wg_map) {
    __shared__ int l_cpre[SL_MAX_BUCKETS + 2];
    __shared__ int wsum[SL_WAVES];
    int cnt2[2];
    int v = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int b = 2 * threadIdx.x + k;
        cnt2[k] = 0;
        if (b <= nb) {
            const uint32_t s = rstart[b];
            uint32_t c = 0;
            if (b < nb) { const uint32_t cap = rstart[b + 1] - s; const uint32_t cu = rcur[(size_t)b * CS_CUR_STRIDE]; c = cu < cap ? cu : cap; }
            bstart[b] = s; bend[b] = s + c;
            if (b == nb) { bstart[nb + 1] = s; bend[nb + 1] = s; }
            if (b < nb) cnt2[k] = (int)((c + (uint32_t)jchunk - 1u) / (uint32_t)jchunk);
        }
        v += cnt2[k];
    }
    int total;
    const int pre = sl_block_exclusive_sum(v, wsum, &total);
    if (2 * threadIdx.x <= nb) l_cpre[2 * threadIdx.x] = pre;
    if (2 * threadIdx.x + 1 <= nb) l_cpre[2 * threadIdx.x + 1] = pre + cnt2[0];
    if (threadIdx.x == 0) { meta[0] = total; l_cpre[nb + 1] = total; }
    __syncthreads();
    for (int w = threadIdx.x; w < total; w += SL_THREADS) {
        int lo = 0, hi = nb;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (l_cpre[m + 1] <= w) lo = m + 1; else hi = m; }
        wg_map[w] = make_int2(lo, w - l_cpre[lo]);
    }
}

// ---- partition, pass 2, STABLE variant (the deterministic count -> fill pair) ---------------------------------------------------
// As k_slice_scatter: rank inside (wavefront, bucket) from match-any ballots against the wavefront's private counter row, prefix
// over the wavefronts per bucket: the records of a bucket keep their input order, whatever the timing of the wavefronts.
struct CsPartSLds { int cm, cell, spl, base, lstart, tot, wcnt, rs, re, rr, d, wsum, total; };
__host__ __device__ inline CsPartSLds cs_part_s_lds(int nb, int ncells, int n_contigs) {
    CsPartSLds L;
    const int nbp = (nb + 2 + 1) & ~1;                          // counters per wavefront row, even
    int o = 0;
    L.cm = o; o += 16 * ((n_contigs + 3) & ~3);
    L.spl = o; o += 8 * nb;
    L.cell = o; o += 4 * ncells;
    L.rs = (o + 15) & ~15; o = L.rs + 4 * CS_TILE;
    L.re = o; o += 4 * CS_TILE;
    L.rr = o; o += 4 * CS_TILE;
    L.base = o; o += 4 * (nb + 2);
    L.lstart = o; o += 4 * (nb + 2);
    L.tot = o; o += 4 * (nb + 2);
    L.wcnt = (o + 3) & ~3; o = L.wcnt + 2 * nbp * CS_WAVES;
    L.d = (o + 3) & ~3; o = L.d + 2 * CS_TILE;
    L.wsum = (o + 3) & ~3; o = L.wsum + 4 * CS_WAVES;
    L.total = o;
    return L;
}

template <bool STRICT>
__global__ __launch_bounds__(CS_THREADS) void k_cs_scatter_stable(CsTab tab, CsGeom g, int nbits, const int32_t* __restrict__ pc, const int32_t* __restrict__ ps,
                                                                 const int32_t* __restrict__ pe, const int32_t* __restrict__ row_id, int64_t n,
                                                                 int chunk, int nchunks, const uint32_t* __restrict__ blk_off,
                                                                 int32_t* __restrict__ out /* 3 int32 per record */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
    const CsPartSLds L = cs_part_s_lds(g.nb, g.ncells, g.n_contigs);
    unsigned long long* l_spl = reinterpret_cast<unsigned long long*>(cs_lds + L.spl);
    int4* l_cm = reinterpret_cast<int4*>(cs_lds + L.cm);
    uint32_t* l_cell = reinterpret_cast<uint32_t*>(cs_lds + L.cell);
    int32_t* l_rs = reinterpret_cast<int32_t*>(cs_lds + L.rs);
    int32_t* l_re = reinterpret_cast<int32_t*>(cs_lds + L.re);
    int32_t* l_rr = reinterpret_cast<int32_t*>(cs_lds + L.rr);
    uint32_t* base = reinterpret_cast<uint32_t*>(cs_lds + L.base);
    uint32_t* lstart = reinterpret_cast<uint32_t*>(cs_lds + L.lstart);
    uint32_t* tot = reinterpret_cast<uint32_t*>(cs_lds + L.tot);
    unsigned short* wcnt = reinterpret_cast<unsigned short*>(cs_lds + L.wcnt);
    unsigned short* l_d = reinterpret_cast<unsigned short*>(cs_lds + L.d);
    uint32_t* wsum = reinterpret_cast<uint32_t*>(cs_lds + L.wsum);
    const int nbp = (g.nb + 2 + 1) & ~1;
    const int tid = threadIdx.x, w = tid / kWave, lane = tid & (kWave - 1);
    const int nbk = g.nb + 1;                                   // buckets incl. the "no candidate" one
    cs_load_tab(tab, g, l_spl, l_cm, l_cell, CS_THREADS);
    for (int k = tid; k < nbk; k += CS_THREADS) { base[k] = blk_off[(int64_t)k * nchunks + blockIdx.x]; tot[k] = 0; }
    for (int k = tid; k < nbp * CS_WAVES / 2; k += CS_THREADS) reinterpret_cast<uint32_t*>(wcnt)[k] = 0;
    __syncthreads();
    const int64_t cbase = (int64_t)blockIdx.x * chunk;
    const int64_t cend = cbase + chunk < n ? cbase + chunk : n;
    const uint64_t lt = lanemask_lt();
    unsigned short* my = wcnt + w * nbp;
    // wavefront w owns the tile elements [w * 256, (w + 1) * 256): item j of lane l = w * 256 + j * 64 + l (input order)
    const int el0 = w * (CS_ITEMS * kWave) + lane;
    for (int64_t tbase = cbase; tbase < cend; tbase += CS_TILE) {
        const int tile_n = (int)((cend - tbase) < (int64_t)CS_TILE ? (cend - tbase) : (int64_t)CS_TILE);
        int32_t s[CS_ITEMS], e[CS_ITEMS], r[CS_ITEMS];
        uint32_t d[CS_ITEMS], rank[CS_ITEMS];
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            const int64_t i = tbase + el0 + j * kWave;
            const bool valid = el0 + j * kWave < tile_n;
            const int32_t c = valid ? __builtin_nontemporal_load(pc + i) : -1;
            s[j] = valid ? __builtin_nontemporal_load(ps + i) : 0;
            e[j] = valid ? __builtin_nontemporal_load(pe + i) : 0;
            r[j] = valid ? (row_id ? __builtin_nontemporal_load(row_id + i) : (int32_t)i) : -1;
            d[j] = !valid ? 0u : cs_bucket<STRICT>(l_spl, l_cm, l_cell, g.nb, g.n_contigs, c, e[j]);
        }
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            const bool valid = el0 + j * kWave < tile_n;
            const uint64_t peers = wave_match_n(d[j], valid, nbits);
            const uint32_t rk = (uint32_t)__popcll(peers & lt);
            const uint32_t before = valid ? (uint32_t)my[d[j]] : 0u;
            rank[j] = before + rk;
            __builtin_amdgcn_wave_barrier();
            if (valid && rk == 0) my[d[j]] = (unsigned short)(before + (uint32_t)__popcll(peers));
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();                                                        // (A) all wavefront rows counted
        uint32_t x0 = 0, x1 = 0;
        {
            const int b0 = 2 * tid;
            if (b0 < nbk) {
                base[b0] += tot[b0];
                if (b0 + 1 < nbk) base[b0 + 1] += tot[b0 + 1];
                uint32_t* row32 = reinterpret_cast<uint32_t*>(wcnt) + tid;
#pragma unroll
                for (int k = 0; k < CS_WAVES; ++k) {
                    const uint32_t v = row32[k * (nbp / 2)];
                    row32[k * (nbp / 2)] = x0 | (x1 << 16);
                    x0 += v & 0xffffu; x1 += v >> 16;
                }
                tot[b0] = x0;
                if (b0 + 1 < nbk) tot[b0 + 1] = x1;
            }
        }
        uint32_t tsum;
        const uint32_t pre = sl_block_exclusive_sum(x0 + x1, wsum, &tsum);      // (B), (C)
        if (2 * tid < nbk) { lstart[2 * tid] = pre; if (2 * tid + 1 < nbk) lstart[2 * tid + 1] = pre + x0; }
        __syncthreads();                                                        // (D)
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            if (el0 + j * kWave < tile_n) {
                const uint32_t pos = lstart[d[j]] + (uint32_t)my[d[j]] + rank[j];
                l_rs[pos] = s[j]; l_re[pos] = e[j]; l_rr[pos] = r[j];
                l_d[pos] = (unsigned short)d[j];
            }
        }
        __syncthreads();                                                        // (E) tile sorted in LDS
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            const int il = j * CS_THREADS + tid;
            if (il < tile_n) {
                const uint32_t dd = l_d[il];
                cs_rec v; v.x = l_rs[il]; v.y = l_re[il]; v.z = l_rr[il];
                *reinterpret_cast<cs_rec*>(out + 3 * (int64_t)(base[dd] + ((uint32_t)il - lstart[dd]))) = v;
            }
        }
        for (int k = tid; k < nbp * CS_WAVES / 2; k += CS_THREADS) reinterpret_cast<uint32_t*>(wcnt)[k] = 0;
        __syncthreads();                                                        // (F) counters clear, staging free
    }
}

// ---- the join ---------------------------------------------------------------------------------------------------------------------
// FUSED: one pass, every tile reserves its output range with one atomic (tile order not reproducible).
// COUNT / FILL: the deterministic pair -- COUNT writes the pairs of every (tile, wavefront) to its own slot, the host scans the
// slots, FILL matches again and every wavefront emits at ITS scanned base: no atomics, no cross-wavefront hand-off, and (with the
// stable partition) an output that is identical from run to run.
enum { CS_FUSED = 0, CS_COUNT = 1, CS_FILL = 2 };

// What COUNT knows about a probe and k_cs_fill needs to emit its pairs without matching again: one word per probe record.
//   bits 0 .. 12   first examined row of the slice (al; at most SL_MAX_ROWS - 1) -- or, flagged by k_cs_join_plain, the hi-bound
//   bits 13 .. 28  match mask of the sixteen-row window (bit t <=> row al + t)
//   bit 31         the window runs on below the examined rows: k_cs_fill redoes that part (plain: recounts from hi; walk: walks
//                  the block maxima below al) -- rare by construction of the two kernels' domains
constexpr uint32_t CS_CACHE_FLAG = 0x80000000u;
static_assert(SL_MAX_ROWS <= (1 << 13), "a slice-local row must fit 13 bits of the cache word");
__device__ __forceinline__ uint32_t cs_cache_word(int al, uint32_t mask, bool flagged) {
    return (uint32_t)al | (mask << 13) | (flagged ? CS_CACHE_FLAG : 0u);
}
// the word travels with the probe's row: k_cs_fill then reads 8 bytes per probe instead of the 12-byte record + the word
typedef unsigned int cs_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cs_cache_store(uint2* p, uint32_t word, int32_t qrow) {
    cs_u2 v; v.x = word; v.y = (uint32_t)qrow;
    __builtin_nontemporal_store(v, reinterpret_cast<cs_u2*>(p));
}
__device__ __forceinline__ uint2 cs_cache_load(const uint2* p) {
    const cs_u2 v = __builtin_nontemporal_load(reinterpret_cast<const cs_u2*>(p));
    return make_uint2(v.x, v.y);
}

struct CsJoinArgs {
    const int32_t* b_start;
    const int2* ep;
    const int32_t* b_row;
    const unsigned short* bins;       // per-slice start bins (k_cs_bins)
    const int4* smeta;                // per-slice metadata (two int4)
    HierView hier;                    // the sorted ends and their block maxima (index_view.hip.h)
    const int32_t* rec;               // bucket-ordered probe records {start, end, row}
    const uint32_t* bstart;           // nb + 2 bucket starts
    const uint32_t* bend;             // SAMPLED partition: end of every bucket's records inside its region (nullptr: bstart[k + 1])
    const int32_t* meta;              // [0] = number of join workgroups
    const int2* wg_map;               // workgroup -> (bucket, chunk inside the bucket)
    int R;
    int jchunk;                       // probes per join workgroup (multiple of the tile)
    int wcap;                         // staging entries per wavefront
    int ablate;                       // profiling only (IVJ_SLICE_ABLATE): 32 no copy-out
    long long capacity;
    unsigned long long* state;        // FUSED: [0] cursor, [1] overflow flag
    long long* wslot;                 // COUNT: pairs per (tile, wavefront), written; FILL: their exclusive scan, read
    uint2* cache;                     // COUNT -> k_cs_fill: {cs_cache_word, probe row} per probe record, bucket order
    int32_t* out_probe;
    int32_t* out_build;
    uint32_t* hw;                     // FUSED: host words [8..12] {pairs, flags, seq}, written by the last workgroup to finish (nullptr: the host copies the state)
    uint32_t hw_seq;
    uint32_t* done;                   // FUSED: finished workgroups (cleared with the state)
    uint32_t* cursor;                 // persistent workgroups (k_cs_join_plain): items drawn from each XCD's part of the chunk list, zeroed per launch (nullptr: one item per workgroup)
    int pmax, pgrain;                 // items per draw: clamp(items left in the XCD's part / pgrain, 1, pmax)
    unsigned long long* trace;        // IVJ_CS_WGTRACE (diagnosis): {hw id | xcc << 32, start, slice staged, end, next run's preparation: start, end} per run, 100-MHz clock
};

// workgroup time line of the plain join (tools/wgtrace.py): who ran where and when -- tail, gaps between workgroups, slice staging share
__device__ __forceinline__ void cs_trace(unsigned long long* trace, int v, int slot) {
    if (!trace || threadIdx.x != 0) return;                                    // (uniform on the pointer)
    if (slot == 0) trace[6 * (size_t)v] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
    trace[6 * (size_t)v + 1 + slot] = wall_clock64();
}

// FUSED: the last join workgroup to finish hands {pairs, flags} to the host words.  Every wavefront first waits for its own memory
// operations (the tile cursors' atomics on the state words among them), the workgroup meets, one thread takes a ticket; the holder
// of the last ticket reads the state words past the caches.  A wavefront that left on a protocol timeout has waited for its flag
// (IVJ_TILE_WAIT); a workgroup all of whose wavefronts left never takes a ticket, the sequence number stays behind and the host copies.
__device__ __forceinline__ void cs_publish_state(uint32_t* hw, uint32_t hw_seq, uint32_t* done, unsigned long long* state, int total_wg) {
    if (!hw) return;                                                           // uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(done, 1u);
        if (t == (uint32_t)total_wg - 1u) {
            const unsigned long long pairs = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long flags = __hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hw_store(hw + 8, (uint32_t)pairs); hw_store(hw + 9, (uint32_t)(pairs >> 32));
            hw_store(hw + 10, (uint32_t)flags); hw_store(hw + 11, (uint32_t)(flags >> 32));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            hw_store(hw + 12, hw_seq);
        }
    }
}
__device__ __forceinline__ void cs_publish_state(const CsJoinArgs& A, int total_wg) { cs_publish_state(A.hw, A.hw_seq, A.done, A.state, total_wg); }

// Copy-out of a wavefront's staged pairs (round 5): the output range starts at an arbitrary element of the two result columns, so a
// plain `for (i = lane; ...)` makes EVERY 256-byte store instruction straddle five 64-byte lines (round-4 counters: 32.1 M write
// requests for 24.8 M lines of pairs).  The loop starts at lane - (misalignment in elements) instead: the first trip writes the
// partial head line(s), every later one four whole lines.  IVJ_SLICE_ABLATE bit 8192 keeps the unaligned form (A/B runs).
constexpr int CS_ABLATE_UNALIGNED = 8192;
__device__ __forceinline__ int cs_copy_align(const int32_t* op, int ablate) {
    return (ablate & CS_ABLATE_UNALIGNED) ? 0 : (int)((reinterpret_cast<uintptr_t>(op) >> 2) & 15u);
}

// Copy-out of a wavefront's staged pairs with SIXTEEN bytes per lane and store (round 6).  A wavefront's output range starts and ends inside
// 64-byte lines it shares with its neighbours' ranges, and what those shared lines cost depends on the store width (tools/micro/write_bw.hip,
// profiles/r06/write_bw_by_store_width.txt: ranges of 508 elements to two columns reach 3.0 TB/s with 4-byte stores, 3.9 with 16-byte ones;
// ranges of whole lines 6.0).  Quads of elements aligned to 16 bytes in global memory: quad q = elements [4 q - E, 4 q - E + 4), E = the
// range's element offset inside its first line; the (at most three) elements of a quad cut by either end of the range go one by one.
// Entry = probe slot << SLOT_SHIFT | row; rows[] may be an LDS pointer with a bias folded in.  k_cs_fill: 0.70 -> 0.61 ms on config 3; the
// fused joins (not bound by their stores: k_cs_join_plain 0.818 / 0.810 against 0.810 / 0.815 ms, k_cs_join 0.226 against 0.224) keep the 4-byte form.  IVJ_SLICE_ABLATE bit 16384: 4-byte stores (A/B, exact).
constexpr int CS_ABLATE_STORE4 = 16384;
template <int SLOT_SHIFT, class RowPtr>
__device__ __forceinline__ void cs_copy_pairs16(const uint32_t* stw, const int32_t* qrw, RowPtr rows, int32_t* op, int32_t* ob, int n, int lane) {
    constexpr uint32_t RM = (1u << SLOT_SHIFT) - 1u;
    const int E = (int)((reinterpret_cast<uintptr_t>(op) >> 2) & 15u);
    typedef int cs_v4 __attribute__((ext_vector_type(4)));
    for (int i0 = 4 * lane - E; i0 < n; i0 += 4 * kWave) {
        if (i0 >= 0 && i0 + 4 <= n) {
            const uint32_t e0 = stw[i0], e1 = stw[i0 + 1], e2 = stw[i0 + 2], e3 = stw[i0 + 3];
            cs_v4 vp, vb;
            vp.x = qrw[e0 >> SLOT_SHIFT]; vp.y = qrw[e1 >> SLOT_SHIFT]; vp.z = qrw[e2 >> SLOT_SHIFT]; vp.w = qrw[e3 >> SLOT_SHIFT];
            vb.x = (int32_t)rows[e0 & RM]; vb.y = (int32_t)rows[e1 & RM]; vb.z = (int32_t)rows[e2 & RM]; vb.w = (int32_t)rows[e3 & RM];
            __builtin_nontemporal_store(vp, reinterpret_cast<cs_v4*>(op + i0));
            __builtin_nontemporal_store(vb, reinterpret_cast<cs_v4*>(ob + i0));
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                if ((unsigned)i < (unsigned)n) {
                    const uint32_t e = stw[i];
                    __builtin_nontemporal_store(qrw[e >> SLOT_SHIFT], op + i);
                    __builtin_nontemporal_store((int32_t)rows[e & RM], ob + i);
                }
            }
        }
    }
}

constexpr int CS_ARGS_LDS = 512;
struct CsJoinLds { int end, pmx, start, row, bin, qrow, stage, ctl, args, total; };
__host__ __device__ inline CsJoinLds cs_join_lds(int R, int wcap) {
    CsJoinLds L;
    int o = 0;
    L.end = o; o += 4 * (R + 16);                      // sixteen ends are read from any aligned row <= rk
    L.pmx = o; o += 4 * (R + 4);                       // [0] = prefix max below the slice, [i + 1] = prefix max of row i
    L.start = o; o += 4 * (R + 4);                     // four pad rows (INT32_MAX) behind the last one
    L.row = o; o += 4 * R;
    L.bin = o; o += 2 * (2 * R + 8);
    L.qrow = (o + 15) & ~15; o = L.qrow + 4 * CS_TILE;
    L.stage = o; o += 4 * wcap * CS_WAVES;
    L.ctl = (o + 15) & ~15; o = L.ctl + 160;               // tile control blocks (64 bytes) + the run loop: group in hand (32), next run (64)
    L.args = o; o += CS_ARGS_LDS;                          // k_cs_join_plain: the kernel arguments its cold code reads
    L.total = o;
    return L;
}

// ---- the join kernel for build sides WITHOUT a tail of long rows (cs_far_share below CS_FAR_LIMIT: the synthetic configs, exons) ----
// The same tiles, LDS layout and emission as k_cs_join below, but a probe whose window runs on below the branch-free one is
// recounted row by row and its wavefront writes from the lanes -- no walk over the block maxima, no staging of rows below the
// slice.  It is kept as a kernel of its own because the walk, inlined into the hot loops, costs the benign case 4 % of the join
// (0.91 -> 0.95 ms) and the index 0.03 ms for the maxima; the host picks the kernel per index from ONE statistic of the build
// side (host_cslice.hip.h::cs_far_share).  Exact on every input; slow (row-by-row) only where k_cs_join is the one chosen.
template <bool STRICT, int MODE>
__global__ __launch_bounds__(CS_THREADS) void k_cs_join_plain(CsJoinArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
    // Arguments the tile loops do not need are read from a COPY IN LDS at their (cold) place of use: pointers loaded at the kernel's entry
    // stay in scalar registers for its whole life, and with the run loop's state the kernel needed 106 of them -- the compiler parked the
    // excess in vector-register lanes (226 v_readlane in the listing, the join 4 % slower).  (Re-reading the kernel-argument segment
    // itself was worse: it lives in host memory, every miss of the scalar cache is a trip over the bus -- the join 0.91 -> 1.0 ms.)
    static_assert(sizeof(CsJoinArgs) <= CS_ARGS_LDS && sizeof(CsJoinArgs) % 4 == 0, "LDS copy of the arguments");
    typedef const __attribute__((address_space(3))) CsJoinArgs* cs_largs_t;
    const CsJoinLds L = cs_join_lds(A.R, A.wcap);
    auto ca = [&]() { return (cs_largs_t)(cs_lds + L.args); };
    int32_t* l_end = reinterpret_cast<int32_t*>(cs_lds + L.end);
    int32_t* l_pmx = reinterpret_cast<int32_t*>(cs_lds + L.pmx);
    int32_t* l_start = reinterpret_cast<int32_t*>(cs_lds + L.start);
    int32_t* l_row = reinterpret_cast<int32_t*>(cs_lds + L.row);
    unsigned short* l_bin = reinterpret_cast<unsigned short*>(cs_lds + L.bin);
    int32_t* l_qrow = reinterpret_cast<int32_t*>(cs_lds + L.qrow);
    uint32_t* l_stage = reinterpret_cast<uint32_t*>(cs_lds + L.stage);
    unsigned long long* lc = reinterpret_cast<unsigned long long*>(cs_lds + L.ctl);     // [2][2] {cursor, base}
    int* li = reinterpret_cast<int*>(lc + 4);                                           // [2][4] {arrived, done, ready, seq}

    // The chunk list {(bucket, chunk of jchunk probes)} is cut into eight contiguous parts, one per XCD (the slices of neighbouring
    // buckets meet in one L2).  Round 6: the workgroups are PERSISTENT (A.cursor; one per CU, grid = CUs): a workgroup draws GROUPS of
    // consecutive list items from its XCD's cursor -- up to A.pmax at a time while the list is long, single items towards its end, and
    // from the other XCDs' lists once its own is empty -- and joins consecutive items of one bucket as ONE run over the slice it staged
    // once (tools/wgtrace.py, profiles/r06/wgtrace_*.txt: with one workgroup per item 8.2 % of the CU time passed between workgroups
    // or behind the last one, and 9.3 % of a workgroup's time was the staging of its slice).  A.cursor = nullptr: one item per
    // workgroup in launch order, workgroup b on XCD b % 8 (rounds 3-5; IVJ_CS_PERSIST=0).
    {
        const __attribute__((address_space(4))) uint32_t* kp = (const __attribute__((address_space(4))) uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();
        if (threadIdx.x < sizeof(CsJoinArgs) / 4) reinterpret_cast<uint32_t*>(cs_lds + L.args)[threadIdx.x] = kp[threadIdx.x];   // (A is the only parameter: offset 0)
        __syncthreads();
    }
    // (the thread index is made opaque once per run: the per-lane addresses derived from it -- staging regions, LDS slots of the items --
    // are otherwise invariants of the run loop, kept in registers across the slice staging, and spilled: 60 bytes of scratch per lane)
    int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
    // the run in hand (uniform): list items [v, v + nch) = probes [q0, q1) of bucket k, whose slice is in LDS
    int v = 0, k = -1;
    int64_t q0 = 0, q1 = 0;
    int32_t smin = 0;
    int bshift = 0, ncell = 1, seg_a = 0, rk = 0, r0 = 0;
    // slice k = sorted rows [r0, r0 + rk): ends / prefix maxima / starts / build rows / bins -> LDS.  Every global load of the slice is
    // requested before the first LDS store (two row quads and six bin words per thread at most): one memory round trip instead of three
    typedef int v4u __attribute__((ext_vector_type(4), aligned(4)));             // 16-byte global loads at any 4-byte aligned row
    static_assert(SL_MAX_ROWS <= 2 * CS_THREADS * 4 && SL_MAX_ROWS + 1 <= 6 * CS_THREADS, "trip counts of the slice staging");
    int32_t pm0 = 0;                                                             // prefix max of the row below the slice
    auto stage_slice = [&]() {
        const cs_largs_t P = ca();
        // 2 R + 2 bins per slice: the stride is even, so pairs of bins are 4-byte aligned
        const uint32_t* gb32 = reinterpret_cast<const uint32_t*>(P->bins + (size_t)k * (size_t)(2 * P->R + CS_BIN_STRIDE_PAD));
        uint32_t* lb32 = reinterpret_cast<uint32_t*>(l_bin);
        const int nb32 = (ncell + 2) / 2;
        const int32_t* gs = P->b_start + r0;
        const int32_t* gr = P->b_row + r0;
        const int32_t* ge = reinterpret_cast<const int32_t*>(P->ep + r0);
        // rows 0 .. 4095 (every thread one quad) and the bins travel together; rows 4096 .. (a quarter of the threads) behind them; the
        // (up to three) rows beyond the last whole quad one by one
        const int rk4 = rk & ~3;
        auto put_quad = [&](int i, const v4u& s4, const v4u& r4, const v4u& e01, const v4u& e23) {
            *reinterpret_cast<int4*>(l_start + i) = make_int4(s4.x, s4.y, s4.z, s4.w);
            *reinterpret_cast<int4*>(l_row + i) = make_int4(r4.x, r4.y, r4.z, r4.w);
            *reinterpret_cast<int4*>(l_end + i) = make_int4(e01.x, e01.z, e23.x, e23.z);
            l_pmx[i + 1] = e01.y; l_pmx[i + 2] = e01.w; l_pmx[i + 3] = e23.y; l_pmx[i + 4] = e23.w;
        };
        {
            const int i = tid * 4;
            v4u s4 = {0, 0, 0, 0}, r4 = s4, e01 = s4, e23 = s4;
            if (i < rk4) {
                s4 = *reinterpret_cast<const v4u*>(gs + i); r4 = *reinterpret_cast<const v4u*>(gr + i);
                e01 = *reinterpret_cast<const v4u*>(ge + 2 * i); e23 = *reinterpret_cast<const v4u*>(ge + 2 * i + 4);
            }
            uint32_t bw[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) { const int i2 = tid + u * CS_THREADS; bw[u] = i2 < nb32 ? gb32[i2] : 0u; }
            if (i < rk4) put_quad(i, s4, r4, e01, e23);
#pragma unroll
            for (int u = 0; u < 6; ++u) { const int i2 = tid + u * CS_THREADS; if (i2 < nb32) lb32[i2] = bw[u]; }
        }
        {
            const int i = tid * 4 + CS_THREADS * 4;
            if (i < rk4) {
                const v4u s4 = *reinterpret_cast<const v4u*>(gs + i), r4 = *reinterpret_cast<const v4u*>(gr + i);
                const v4u e01 = *reinterpret_cast<const v4u*>(ge + 2 * i), e23 = *reinterpret_cast<const v4u*>(ge + 2 * i + 4);
                put_quad(i, s4, r4, e01, e23);
            }
        }
        if (tid < (rk & 3)) {
            const int j = rk4 + tid;
            l_start[j] = gs[j]; l_row[j] = gr[j];
            l_end[j] = ge[2 * j]; l_pmx[j + 1] = ge[2 * j + 1];
        }
        if (tid < 4) l_start[rk + tid] = INT32_MAX;
        if (tid < 16) l_end[rk + tid] = INT32_MIN;
        if (tid == 0) l_pmx[0] = pm0;
    };

    uint32_t* stw = l_stage + wv * A.wcap;                                     // this wavefront's staging entries
    int32_t* qrw = l_qrow + wv * CS_WTILE;                                     // this wavefront's probe rows of the tile
    auto ld = [](const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto stv = [](int* p, int x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };

    // records of the tile: wavefront wv owns the contiguous probes [wv * 256, (wv + 1) * 256), item j of lane l = wv * 256 + j * 64 + l
    // (lb > 0: 8-byte records {(end - slice minimum) << lb | (end - start), row}, unpacked in match_tile; uniform)
    const int lb = __builtin_amdgcn_readfirstlane(ca()->meta[CS_META_FMT]);
    // The run loop's own state lives in LDS, not in scalar registers (the tile loops need every one of those), and run i + 1 is prepared by
    // the FIRST WAVEFRONT TO FINISH run i, while the others still work on their last tiles: drawing a group from the list cursors, the
    // items' (bucket, chunk) entries, the bucket's bounds and the slice's metadata are a chain of four dependent memory round trips -- in
    // front of every run they cost 5 us of an idle CU, and on one wavefront at the START of a run they delay every tile of it (a tile's
    // output range is reserved when its LAST wavefront arrives).
    //   l_item = {first item of the group in hand, its items, items already taken, lists given up | a draw happened << 8, bucket of the
    //            slice in LDS, own cursor after the last draw, wavefronts that finished the run, items of the list} (the preparing lane only);
    //   l_next = the next run: {1 | stage the slice << 1 (-1: no more work, -2: this workgroup never had any), first item, bucket, -,
    //            q0, q1 (64 bits each), the slice's two metadata quads}, written behind barrier (B) of run i, read behind barrier (T) of run i + 1
    int* l_item = li + 8;
    int* l_next = li + 16;
    auto prepare_next = [&]() {                                                 // one lane
        // (three dependent memory round trips, ~ 2 us each under load: the cursor's atomic, the group's list entries at once, the run's
        // bounds and slice metadata at once -- the list length is kept in LDS and a cursor's position is guessed from the last draw)
        const cs_largs_t P = ca();
        int g_v0 = l_item[0], g_cnt = l_item[1], g_i = l_item[2];
        const int2* wm = P->wg_map;
        if (g_i >= g_cnt) {
            int v0 = -1, cnt = 0;
            const int total_wg = l_item[7];
            const int per = (total_wg + 7) / 8;
            int victim = l_item[3] & 255;
            const int drew = l_item[3] >> 8;
            uint32_t* cur = P->cursor;
            if (cur) {
                const int home = (int)(__builtin_amdgcn_s_getreg(63508) & 7u);  // XCC_ID: the list this CU serves first
                const int pgrain = P->pgrain, pmax = P->pmax;
                int seen = victim == 0 ? l_item[5] : 0;                         // own list: at least what the last draw saw
                while (victim < 8) {
                    const int x = (home + victim) & 7;
                    const int lo = x * per;
                    const int len = per < total_wg - lo ? per : total_wg - lo;
                    if (len > 0 && seen < len) {
                        int m = 1;
                        if (victim == 0) { m = (len - seen) / pgrain; m = m < 1 ? 1 : (m > pmax ? pmax : m); }
                        const int t = (int)atomicAdd(cur + x, (uint32_t)m);
                        if (t < len) { v0 = lo + t; cnt = m < len - t ? m : len - t; l_item[5] = t + m; break; }
                    }
                    ++victim; seen = 0;
                }
            } else if (!drew) {
                const int ws = (int)(blockIdx.x >> 3);
                const int vv = (int)(blockIdx.x & 7) * per + ws;
                if (ws < per && vv < total_wg) { v0 = vv; cnt = 1; }
            }
            l_item[0] = v0; l_item[1] = cnt; l_item[3] = victim | ((drew || v0 >= 0) ? 256 : 0);
            g_v0 = v0; g_cnt = cnt; g_i = 0;
            if (v0 < 0) { l_next[0] = (!cur && !drew) ? -2 : -1; return; }
        }
        const int vv = g_v0 + g_i;
        const int left = g_cnt - g_i;
        // the group's consecutive items of one bucket are ONE run (entries read four at a time)
        const int2 bc = wm[vv];
        int nch = 1;
        for (int b0 = 1; b0 < left; b0 += 4) {
            int kx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) kx[u] = b0 + u < left ? wm[vv + b0 + u].x : -1;
            int same = 0;
#pragma unroll
            for (int u = 3; u >= 0; --u) same = kx[u] == bc.x ? same + 1 : 0;   // (leading entries of this bucket)
            nch += same;
            if (same < 4) break;
        }
        l_item[2] = g_i + nch;
        const int jc = P->jchunk;
        const uint32_t* be = P->bend;
        const uint32_t* bs = P->bstart;
        const uint32_t bs0 = bs[bc.x], be0 = be ? be[bc.x] : bs[bc.x + 1];
        const int4 m0 = P->smeta[2 * bc.x], m1 = P->smeta[2 * bc.x + 1];
        const int64_t a0 = (int64_t)bs0 + (int64_t)bc.y * jc;
        const int64_t qe = a0 + (int64_t)nch * jc;
        const int64_t a1 = qe < (int64_t)be0 ? qe : (int64_t)be0;
        l_next[0] = 1 | (bc.x != l_item[4] ? 2 : 0); l_next[1] = vv; l_next[2] = bc.x;
        l_next[4] = (int)(uint32_t)a0; l_next[5] = (int)(a0 >> 32); l_next[6] = (int)(uint32_t)a1; l_next[7] = (int)(a1 >> 32);
        *reinterpret_cast<int4*>(l_next + 8) = m0; *reinterpret_cast<int4*>(l_next + 12) = m1;
        l_item[4] = bc.x;
    };
    auto end_run = [&]() {
        if (lane == 0 && atomicAdd(&l_item[6], 1) == 0) {
            unsigned long long* tr = ca()->trace;
            if (tr) tr[6 * (size_t)v + 4] = wall_clock64();
            prepare_next();
            if (tr) tr[6 * (size_t)v + 5] = wall_clock64();
        }
    };
    if (tid == 0) { l_item[0] = -1; l_item[1] = 0; l_item[2] = 0; l_item[3] = 0; l_item[4] = -1; l_item[5] = 0; l_item[6] = 0; l_item[7] = ca()->meta[0]; }
    __syncthreads();
    for (;;) {
        end_run();                                                              // (the first wavefront to get here prepares the next run -- or the first one)
        __syncthreads();                                                        // (T) every wavefront has left the previous run; l_next is this run
        auto rf = [&](int i) { return __builtin_amdgcn_readfirstlane(l_next[i]); };
        const int flag = rf(0);
        if (flag < 0) {
            if (flag == -2) return;                                             // (a workgroup beyond the list: it never takes a ticket)
            break;
        }
        v = rf(1); k = rf(2);
        q0 = (int64_t)(((uint64_t)(uint32_t)rf(5) << 32) | (uint32_t)rf(4));
        q1 = (int64_t)(((uint64_t)(uint32_t)rf(7) << 32) | (uint32_t)rf(6));
        smin = rf(8); bshift = rf(9); ncell = rf(10); pm0 = rf(11);
        seg_a = rf(12); rk = rf(14); r0 = rf(15);
        cs_trace(ca()->trace, v, 0);
        tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (flag & 2) stage_slice();
        tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                                           // (again: nothing the tile loops derive from it is alive during the staging)
        lane = tid & (kWave - 1); wv = tid / kWave;
        stw = l_stage + wv * A.wcap; qrw = l_qrow + wv * CS_WTILE;
        if (tid < 4) { unsigned long long z = 0; asm volatile("" : "+v"(z)); lc[tid] = z; }   // (made here: hoisted out of the run loop the zero pair was spilled)
        if (tid < 8) li[tid] = (tid == 7) ? 1 : 0;                             // block 1 serves tile 1 first (seq = li[1][3])
        if (tid == 0) l_item[6] = 0;                                            // (wavefronts that have finished the run in hand)
        __syncthreads();                                                        // (B)
        cs_trace(ca()->trace, v, 1);
        // (the tile state is declared per run: at function scope its registers were carried from run to run -- through the staging, which spilled)
    cs_rec nxt[CS_ITEMS];
    auto load_tile = [&](int64_t tb) {
        const int rem = (int)((q1 - tb) < (int64_t)CS_TILE ? (q1 - tb) : (int64_t)CS_TILE);
        if (lb) {
            const int32_t* tp = A.rec + 2 * tb;                                // uniform
#pragma unroll
            for (int j = 0; j < CS_ITEMS; ++j) {
                const int il = wv * CS_WTILE + j * kWave + lane;
                // (the two words land in the record's LOW components: the load then targets the registers itself -- with the row in
                // .z the compiler loaded into a temporary and waited for it at once, which serialised the prefetch: join 0.93 -> 1.11 ms)
                cs_rec8 v; v.x = 0; v.y = -1;
                if (il < rem) v = __builtin_nontemporal_load(reinterpret_cast<const cs_rec8*>(tp + 2 * il));
                nxt[j].x = v.x; nxt[j].y = v.y;
            }
            return;
        }
        const int32_t* tp = A.rec + 3 * tb;                                    // uniform
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            const int il = wv * CS_WTILE + j * kWave + lane;
            if (il < rem) nxt[j] = __builtin_nontemporal_load(reinterpret_cast<const cs_rec*>(tp + 3 * il));
            else { nxt[j].x = 0; nxt[j].y = 0; nxt[j].z = -1; }
        }
    };
    // per-probe state of the tile whose matches are known but not yet emitted
    int32_t qs[CS_ITEMS], qrow[CS_ITEMS];
    int al[CS_ITEMS], hi[CS_ITEMS];                                            // slice-local: first examined row, hi-bound
    uint32_t mask[CS_ITEMS];                                                   // bit t <=> row al + t matches
    int cnt[CS_ITEMS];
    bool lng[CS_ITEMS];
    bool any_lng = false;

    auto match_tile = [&](int64_t tb) {
        int32_t qe[CS_ITEMS];
        bool valid[CS_ITEMS];
        const int rem = (int)((q1 - tb) < (int64_t)CS_TILE ? (q1 - tb) : (int64_t)CS_TILE);
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            qs[j] = nxt[j].x; qe[j] = nxt[j].y; qrow[j] = nxt[j].z;
            valid[j] = wv * CS_WTILE + j * kWave + lane < rem;
        }
        if (lb) {                                                              // uniform: unpack the 8-byte form {packed word, row}
            const uint32_t lmask = (1u << lb) - 1u;
#pragma unroll
            for (int j = 0; j < CS_ITEMS; ++j) {
                const uint32_t w0 = (uint32_t)qs[j];
                qrow[j] = qe[j];
                qe[j] = (int32_t)((uint32_t)smin + (w0 >> lb));
                qs[j] = (int32_t)((uint32_t)qe[j] - (w0 & lmask));
            }
        }
        if (tb + CS_TILE < q1) load_tile(tb + CS_TILE);                        // next tile's records in flight
        // hi-bound: bin of the end, then the (at most four) rows of the bin that start below it.  Rows before the bin's first
        // row start below the bin's lower edge <= end; rows of later bins start above the end: no range checks needed.
        int first[CS_ITEMS];
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            uint32_t cl = ((uint32_t)qe[j] - (uint32_t)smin) >> bshift;
            cl = cl < (uint32_t)(ncell - 1) ? cl : (uint32_t)(ncell - 1);
            first[j] = l_bin[cl];
        }
        int32_t s4[CS_ITEMS][4];
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            const int32_t* sp = l_start + first[j];
            s4[j][0] = sp[0]; s4[j][1] = sp[1]; s4[j][2] = sp[2]; s4[j][3] = sp[3];
        }
        unsigned long long more = 0;
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            unsigned long long m4;
            hi[j] = cs_count4<STRICT>(first[j], s4[j][0], s4[j][1], s4[j][2], s4[j][3], qe[j], &m4);
            more |= m4;
        }
        if (__builtin_expect(more != 0, 0)) {                                  // uniform: some probe may have more than four rows of its bin below its end
#pragma unroll
            for (int j = 0; j < CS_ITEMS; ++j) {
                if (valid[j] && hi[j] == first[j] + 4) {
                    uint32_t cl = ((uint32_t)qe[j] - (uint32_t)smin) >> bshift;
                    cl = cl < (uint32_t)(ncell - 1) ? cl : (uint32_t)(ncell - 1);
                    int lo = hi[j], hh = l_bin[cl + 1];                        // first row of [hi, rend) that does not start below the end
                    while (lo < hh) { const int mid = (lo + hh) >> 1; if (lt_op<STRICT>(l_start[mid], qe[j])) lo = mid + 1; else hh = mid; }
                    hi[j] = lo;
                }
            }
        }
        // window below hi, branch-free: sixteen ends from the 16-byte aligned row at or below hi - CS_WIN (never below row 0);
        // bit t <=> row al + t, rows at or above hi are cut off.  One prefix-max read of the row below the examined ones
        // (l_pmx[0] = the row below the slice, INT32_MIN when the contig starts here) tells whether the window runs on: those
        // probes are redone exactly.
        unsigned long long lmask = 0;
#pragma unroll
        for (int jj = 0; jj < CS_ITEMS; jj += 2) {
            int4 w[2][4];
            int32_t pm[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = jj + u;
                const int h = hi[j] < rk ? hi[j] : rk;
                hi[j] = h;
                int a0 = (h - CS_WIN) & ~3;
                a0 = a0 > 0 ? a0 : 0;
                al[j] = a0;
                const int4* p4 = reinterpret_cast<const int4*>(l_end + a0);
                w[u][0] = p4[0]; w[u][1] = p4[1]; w[u][2] = p4[2]; w[u][3] = p4[3];
                pm[u] = l_pmx[a0];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = jj + u;
                uint32_t m = ends_mask16<STRICT>(qs[j], w[u][0], w[u][1], w[u][2], w[u][3]);
                m = __builtin_amdgcn_ubfe(m, 0u, (uint32_t)(hi[j] - al[j]));    // rows al .. hi - 1
                lng[j] = valid[j] && lt_op<STRICT>(qs[j], pm[u]);
                mask[j] = valid[j] ? m : 0u;
                cnt[j] = __popc(mask[j]);
                lmask |= __ballot(lng[j]);
            }
        }
        any_lng = lmask != 0;
        if (any_lng) {
#pragma unroll
            for (int j = 0; j < CS_ITEMS; ++j) {
                if (lng[j]) {
                    int c2 = 0;
                    for (int p = r0 + hi[j] - 1; p >= seg_a; --p) {
                        const int i = p - r0;
                        int2 vv = make_int2(l_end[i < 0 ? 0 : i], l_pmx[i < 0 ? 1 : i + 1]);
                        if (i < 0) vv = ca()->ep[p];
                        if (!lt_op<STRICT>(qs[j], vv.y)) break;
                        c2 += lt_op<STRICT>(qs[j], vv.x) ? 1 : 0;
                    }
                    cnt[j] = c2;
                }
            }
        }
    };


    if constexpr (MODE != CS_FUSED) {
        // deterministic pair: slot of (workgroup v, tile tix, wavefront wv); every wavefront works on its own
        load_tile(q0);
        const int ntile = (int)((q1 - q0 + CS_TILE - 1) / CS_TILE);
        const int tiles_per_chunk = __builtin_amdgcn_readfirstlane(ca()->jchunk) / CS_TILE;
        for (int tix = 0; tix < ntile; ++tix) {
            match_tile(q0 + (int64_t)tix * CS_TILE);
            const long long slot = ((long long)v * tiles_per_chunk + tix) * CS_WAVES + wv;
            int lsum = 0;
#pragma unroll
            for (int j = 0; j < CS_ITEMS; ++j) lsum += cnt[j];
            const int linc = wave_incl_sum_dpp(lsum);
            const int wtot = __builtin_amdgcn_readlane(linc, kWave - 1);
            if constexpr (MODE == CS_COUNT) {
                if (lane == 0) A.wslot[slot] = (long long)wtot;
                if (A.cache) {                                                 // uniform: the matches go to k_cs_fill instead of being redone there
                    const int64_t tb = q0 + (int64_t)tix * CS_TILE;
                    const int rem = (int)((q1 - tb) < (int64_t)CS_TILE ? (q1 - tb) : (int64_t)CS_TILE);
#pragma unroll
                    for (int j = 0; j < CS_ITEMS; ++j) {
                        const int il = wv * CS_WTILE + j * kWave + lane;
                        if (il < rem) cs_cache_store(A.cache + tb + il, lng[j] ? cs_cache_word(hi[j], 0u, true) : cs_cache_word(al[j], mask[j], false), qrow[j]);
                    }
                }
                continue;
            }
            if (wtot == 0) continue;                                           // uniform
            const long long wbase = A.wslot[slot];
            if (wtot <= A.wcap && !any_lng) {
                int off = linc - lsum;
#pragma unroll
                for (int j = 0; j < CS_ITEMS; ++j) {
                    qrw[j * kWave + lane] = qrow[j];
                    uint32_t m = mask[j];
                    const uint32_t ent = ((uint32_t)(j * kWave + lane) << 16) | (uint32_t)al[j];
                    uint32_t* so = stw + off;
                    while (m) {
                        const int t = __builtin_ctz(m);
                        m &= m - 1;
                        so[0] = ent + (uint32_t)t;
                        if (m) { so[1] = ent + (uint32_t)__builtin_ctz(m); m &= m - 1; }
                        so += 2;
                    }
                    off += cnt[j];
                }
                __builtin_amdgcn_wave_barrier();
                int32_t* op = A.out_probe + wbase;
                int32_t* ob = A.out_build + wbase;
                const int ca = cs_copy_align(op, A.ablate);
#pragma unroll 4
                for (int i = lane - ca; i < wtot; i += kWave) {
                    if ((unsigned)i < (unsigned)wtot) {
                        const uint32_t e = stw[i];
                        __builtin_nontemporal_store(qrw[e >> 16], op + i);
                        __builtin_nontemporal_store(l_row[e & 0xffffu], ob + i);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            } else {
                long long off = wbase + (linc - lsum);
#pragma unroll
                for (int j = 0; j < CS_ITEMS; ++j) {
                    if (!lng[j]) {
                        uint32_t m = mask[j];
                        long long o = off;
                        while (m) {
                            const int t = __builtin_ctz(m);
                            m &= m - 1;
                            A.out_probe[o] = qrow[j]; A.out_build[o] = l_row[al[j] + t];
                            ++o;
                        }
                    } else {
                        long long o = off + cnt[j] - 1;
                        for (int p = r0 + hi[j] - 1; o >= off; --p) {
                            const int i = p - r0;
                            int32_t ev = l_end[i < 0 ? 0 : i];
                            if (i < 0) ev = ca()->ep[p].x;
                            if (lt_op<STRICT>(qs[j], ev)) {
                                int32_t rv = l_row[i < 0 ? 0 : i];
                                if (i < 0) rv = ca()->b_row[p];
                                A.out_probe[o] = qrow[j]; A.out_build[o] = rv; --o;
                            }
                        }
                    }
                    off += cnt[j];
                }
            }
        }
        continue;                                                              // (the next run)
    }
    // Barrier-free tile loop (protocol of slice.hip.h's fused mode): a wavefront's pairs of a tile are contiguous in the
    // tile's output range at the offset a returning LDS atomic on the tile's cursor gives it; the LAST wavefront to arrive
    // reserves the range with the one global atomic and publishes the base in LDS; the others look at it one iteration later.
    load_tile(q0);
    const int ntile = (int)((q1 - q0 + CS_TILE - 1) / CS_TILE);
    const int spin_bound = (A.ablate & SL_ABLATE_TILE_FAULT) ? SL_SPIN_BOUND_TEST : SL_SPIN_BOUND;
    const bool tile_fault = (A.ablate & SL_ABLATE_TILE_FAULT) && v == 0;
    int pend_wtot = -1;                                                        // this wavefront's staged pairs of the previous tile
    long long pend_woff = 0;
    // (finishing tile t -- the copy-out of this wavefront's staged pairs at the base the tile's last wavefront reserved -- happens one
    // iteration later, behind the next tile's matching; the loop is written with the last finish peeled off: with one loop of ntile + 1
    // iterations and the matching under a condition the compiler carried every per-tile register from iteration to iteration)
    auto finish_tile = [&](int t) {
        __builtin_amdgcn_wave_barrier();
        int* c = li + (t & 1) * 4;
        unsigned long long* c64 = lc + (t & 1) * 2;
        bool timed_out = false;
        IVJ_TILE_WAIT(ld(c + 2) == 0, spin_bound, A.state, lane, timed_out = true);
        long long tb = (long long)__hip_atomic_load(c64 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (timed_out) tb = -1;
        if (tb >= 0 && pend_wtot > 0 && !(A.ablate & 32)) {
            int32_t* op = A.out_probe + tb + pend_woff;
            int32_t* ob = A.out_build + tb + pend_woff;
            const int ca = cs_copy_align(op, A.ablate);
#pragma unroll 4
            for (int i = lane - ca; i < pend_wtot; i += kWave) {
                if ((unsigned)i < (unsigned)pend_wtot) {
                    const uint32_t e = stw[i];
                    __builtin_nontemporal_store(qrw[e >> 16], op + i);
                    __builtin_nontemporal_store(l_row[e & 0xffffu], ob + i);
                }
            }
        }
        if (lane == 0) {
            if (atomicAdd(c + 1, 1) == CS_WAVES - 1) {                         // last wavefront out: recycle the block for tile t + 2
                __hip_atomic_store(c64, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                stv(c + 0, 0); stv(c + 1, 0); stv(c + 2, 0);
                stv(c + 3, t + 2);
            }
        }
    };
    for (int tix = 0; tix < ntile; ++tix) {
        match_tile(q0 + (int64_t)tix * CS_TILE);
        if (tix > 0) finish_tile(tix - 1);                                     // its base was requested one iteration ago
        int* c = li + (tix & 1) * 4;
        unsigned long long* c64 = lc + (tix & 1) * 2;
        IVJ_TILE_WAIT(ld(c + 3) != tix, spin_bound, A.state, lane, return);       // the block is ours (recycled after tile tix - 2)
        int lsum = 0;
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) lsum += cnt[j];
        const int linc = wave_incl_sum_dpp(lsum);
        const int wtot = __builtin_amdgcn_readlane(linc, kWave - 1);
        long long woff = 0;
        if (lane == 0) {
            woff = (long long)atomicAdd(c64, (unsigned long long)wtot);
            if (atomicAdd(c + 0, 1) == CS_WAVES - 1) {                         // last wavefront in: the tile's total is complete
                const long long total = (long long)__hip_atomic_load(c64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                long long base = 0;
                if (total > 0) {
                    base = (long long)atomicAdd(&A.state[0], (unsigned long long)total);
                    if (base + total > A.capacity) { atomicOr(&A.state[1], 1ull); base = -1; }
                }
                __hip_atomic_store(c64 + 1, (unsigned long long)base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!(tile_fault && tix == 0)) stv(c + 2, 1);
            }
        }
        woff = ((long long)__builtin_amdgcn_readfirstlane((int)(woff >> 32)) << 32) | (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(woff & 0xffffffffll));
        if (wtot > 0 && wtot <= A.wcap && !any_lng) {
            // usual case: entries {wave-local probe slot << 16 | slice-local row} into the wavefront's staging region
            int off = linc - lsum;
#pragma unroll
            for (int j = 0; j < CS_ITEMS; ++j) {
                qrw[j * kWave + lane] = qrow[j];
                uint32_t m = mask[j];
                uint32_t ent = ((uint32_t)(j * kWave + lane) << 16) | (uint32_t)al[j];
                uint32_t* so = stw + off;
                while (m) {                                                    // two matches per trip: half the loop overhead
                    const int t = __builtin_ctz(m);
                    m &= m - 1;
                    so[0] = ent + (uint32_t)t;
                    if (m) { so[1] = ent + (uint32_t)__builtin_ctz(m); m &= m - 1; }
                    so += 2;
                }
                off += cnt[j];
            }
            pend_wtot = wtot;
        } else if (wtot > 0) {
            // dense wavefront or windows running on below the examined rows: wait for the base now, write from the lanes
            bool timed_out = false;
            IVJ_TILE_WAIT(ld(c + 2) == 0, spin_bound, A.state, lane, timed_out = true);
            long long tb = (long long)__hip_atomic_load(c64 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (timed_out) tb = -1;
            if (tb >= 0) {
                long long off = tb + woff + (linc - lsum);
#pragma unroll
                for (int j = 0; j < CS_ITEMS; ++j) {
                    if (!lng[j]) {
                        uint32_t m = mask[j];
                        long long o = off;
                        while (m) {
                            const int t = __builtin_ctz(m);
                            m &= m - 1;
                            A.out_probe[o] = qrow[j]; A.out_build[o] = l_row[al[j] + t];
                            ++o;
                        }
                    } else {
                        long long o = off + cnt[j] - 1;                        // the f-th match from the top of the window owns slot end - 1 - f
                        for (int p = r0 + hi[j] - 1; o >= off; --p) {
                            const int i = p - r0;
                            int32_t ev = l_end[i < 0 ? 0 : i];
                            if (i < 0) ev = ca()->ep[p].x;
                            if (lt_op<STRICT>(qs[j], ev)) {
                                int32_t rv = l_row[i < 0 ? 0 : i];
                                if (i < 0) rv = ca()->b_row[p];
                                A.out_probe[o] = qrow[j]; A.out_build[o] = rv; --o;
                            }
                        }
                    }
                    off += cnt[j];
                }
            }
            pend_wtot = 0;
        } else pend_wtot = 0;
        pend_woff = woff;
    }
    if (ntile > 0) finish_tile(ntile - 1);
    if (ca()->trace) { end_run(); __syncthreads(); cs_trace(ca()->trace, v, 2); }   // (end_run: only the first call of a run prepares)         // (diagnosis runs only: every wavefront has finished its tiles)
    }                                                                          // (runs of this workgroup)
    cs_publish_state(ca()->hw, ca()->hw_seq, ca()->done, A.state, ca()->cursor ? (int)gridDim.x : ca()->meta[0]);
}

template <bool STRICT, int MODE>
__global__ __launch_bounds__(CS_THREADS) void k_cs_join(CsJoinArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
    const CsJoinLds L = cs_join_lds(A.R, A.wcap);
    int32_t* l_end = reinterpret_cast<int32_t*>(cs_lds + L.end);
    int32_t* l_pmx = reinterpret_cast<int32_t*>(cs_lds + L.pmx);
    int32_t* l_start = reinterpret_cast<int32_t*>(cs_lds + L.start);
    int32_t* l_row = reinterpret_cast<int32_t*>(cs_lds + L.row);
    unsigned short* l_bin = reinterpret_cast<unsigned short*>(cs_lds + L.bin);
    int32_t* l_qrow = reinterpret_cast<int32_t*>(cs_lds + L.qrow);
    uint32_t* l_stage = reinterpret_cast<uint32_t*>(cs_lds + L.stage);
    unsigned long long* lc = reinterpret_cast<unsigned long long*>(cs_lds + L.ctl);     // [2][2] {cursor, base}
    int* li = reinterpret_cast<int*>(lc + 4);                                           // [2][4] {arrived, done, ready, seq}

    // XCD-affine order: workgroup b runs on XCD b % 8 (observed); XCD x takes the contiguous eighth of the (bucket, chunk) list
    const int total_wg = A.meta[0];
    const int per = (total_wg + 7) / 8;
    const int wslot = (int)(blockIdx.x >> 3);
    const int v = (int)(blockIdx.x & 7) * per + wslot;
    if (wslot >= per || v >= total_wg) return;                                 // uniform
    const int2 bc = A.wg_map[v];
    const int k = bc.x;
    const int64_t q0 = (int64_t)A.bstart[k] + (int64_t)bc.y * A.jchunk;
    const int64_t qend = A.bend ? (int64_t)A.bend[k] : (int64_t)A.bstart[k + 1];
    const int64_t q1 = q0 + A.jchunk < qend ? q0 + A.jchunk : qend;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;

    const int4 sm0 = A.smeta[2 * k], sm1 = A.smeta[2 * k + 1];
    const int32_t smin = sm0.x;
    const int bshift = sm0.y, ncell = sm0.z;
    const int seg_a = sm1.x, rk = sm1.z, r0 = sm1.w;
    // slice k = sorted rows [r0, r0 + rk): ends / prefix maxima / starts / build rows / bins -> LDS
    typedef int v4u __attribute__((ext_vector_type(4), aligned(4)));             // 16-byte global loads at any 4-byte aligned row
    for (int i = tid * 4; i < rk; i += CS_THREADS * 4) {
        if (i + 4 <= rk) {
            const v4u s4 = *reinterpret_cast<const v4u*>(A.b_start + r0 + i);
            const v4u r4 = *reinterpret_cast<const v4u*>(A.b_row + r0 + i);
            const v4u e01 = *reinterpret_cast<const v4u*>(reinterpret_cast<const int32_t*>(A.ep + r0 + i));
            const v4u e23 = *reinterpret_cast<const v4u*>(reinterpret_cast<const int32_t*>(A.ep + r0 + i + 2));
            *reinterpret_cast<int4*>(l_start + i) = make_int4(s4.x, s4.y, s4.z, s4.w);
            *reinterpret_cast<int4*>(l_row + i) = make_int4(r4.x, r4.y, r4.z, r4.w);
            *reinterpret_cast<int4*>(l_end + i) = make_int4(e01.x, e01.z, e23.x, e23.z);
            l_pmx[i + 1] = e01.y; l_pmx[i + 2] = e01.w; l_pmx[i + 3] = e23.y; l_pmx[i + 4] = e23.w;
        } else {
            for (int j = i; j < rk; ++j) {
                l_start[j] = A.b_start[r0 + j]; l_row[j] = A.b_row[r0 + j];
                const int2 e = A.ep[r0 + j];
                l_end[j] = e.x; l_pmx[j + 1] = e.y;
            }
        }
    }
    if (tid < 4) l_start[rk + tid] = INT32_MAX;
    if (tid < 16) l_end[rk + tid] = INT32_MIN;
    if (tid == 0) l_pmx[0] = sm0.w;
    {
        const unsigned short* gb = A.bins + (size_t)k * (size_t)(2 * A.R + CS_BIN_STRIDE_PAD);
        // 2 R + 2 entries per slice: the stride is even, so pairs of bins are 4-byte aligned
        const uint32_t* gb32 = reinterpret_cast<const uint32_t*>(gb);
        uint32_t* lb32 = reinterpret_cast<uint32_t*>(l_bin);
        for (int i = tid; i < (ncell + 2) / 2; i += CS_THREADS) lb32[i] = gb32[i];
    }
    if (tid < 4) lc[tid] = 0;
    if (tid < 8) li[tid] = (tid == 7) ? 1 : 0;                                 // block 1 serves tile 1 first (seq = li[1][3])
    __syncthreads();

    uint32_t* stw = l_stage + wv * A.wcap;                                     // this wavefront's staging entries
    int32_t* qrw = l_qrow + wv * CS_WTILE;                                     // this wavefront's probe rows of the tile
    auto ld = [](const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto stv = [](int* p, int x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };

    // records of the tile: wavefront wv owns the contiguous probes [wv * 256, (wv + 1) * 256), item j of lane l = wv * 256 + j * 64 + l
    // (lb > 0: 8-byte records {(end - slice minimum) << lb | (end - start), row}, unpacked in match_tile; uniform)
    const int lb = A.meta[CS_META_FMT];
    cs_rec nxt[CS_ITEMS];
    auto load_tile = [&](int64_t tb) {
        const int rem = (int)((q1 - tb) < (int64_t)CS_TILE ? (q1 - tb) : (int64_t)CS_TILE);
        if (lb) {
            const int32_t* tp = A.rec + 2 * tb;                                // uniform
#pragma unroll
            for (int j = 0; j < CS_ITEMS; ++j) {
                const int il = wv * CS_WTILE + j * kWave + lane;
                // (the two words land in the record's LOW components: the load then targets the registers itself -- with the row in
                // .z the compiler loaded into a temporary and waited for it at once, which serialised the prefetch: join 0.93 -> 1.11 ms)
                cs_rec8 v; v.x = 0; v.y = -1;
                if (il < rem) v = __builtin_nontemporal_load(reinterpret_cast<const cs_rec8*>(tp + 2 * il));
                nxt[j].x = v.x; nxt[j].y = v.y;
            }
            return;
        }
        const int32_t* tp = A.rec + 3 * tb;                                    // uniform
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            const int il = wv * CS_WTILE + j * kWave + lane;
            if (il < rem) nxt[j] = __builtin_nontemporal_load(reinterpret_cast<const cs_rec*>(tp + 3 * il));
            else { nxt[j].x = 0; nxt[j].y = 0; nxt[j].z = -1; }
        }
    };
    // per-probe state of the tile whose matches are known but not yet emitted
    int32_t qs[CS_ITEMS], qrow[CS_ITEMS];
    int al[CS_ITEMS], hi[CS_ITEMS];                                            // slice-local: first examined row, hi-bound
    uint32_t mask[CS_ITEMS];                                                   // bit t <=> row al + t matches
    int cnt[CS_ITEMS];
    bool any_lng = false;

    // Matches of a probe starting at qsv among the sorted rows BELOW slice-local row a0 (the rows the branch-free window does not
    // cover; all of them start below the probe's end, so a row matches iff it ends above qsv), visited in descending position;
    // f(p) gets the global sorted position: hier_walk (index_view.hip.h), with the slice's rows read from LDS.
    auto ep_at = [&](int p) -> int2 {
        const int i = p - r0;
        if (i >= 0) return make_int2(l_end[i], l_pmx[i + 1]);
        return A.ep[p];
    };
    auto walk_below = [&](int32_t qsv, int a0, auto&& f) {
        // a window usually runs on by a row or two: those come from the slice in LDS, row by row; only a window that is still
        // open CS_LIN rows further down (or at the slice's first row) takes the block maxima in HBM
        int i = a0 - 1;
        const int stop = a0 - CS_LIN > 0 ? a0 - CS_LIN : 0;
        for (; i >= stop; --i) {
            if (!lt_op<STRICT>(qsv, l_pmx[i + 1])) return;
            if (lt_op<STRICT>(qsv, l_end[i])) { if (!f(r0 + i)) return; }
        }
        if (r0 + i >= seg_a) hier_walk<STRICT>(A.hier, ep_at, seg_a, r0 + i, qsv, f);
    };

    auto match_tile = [&](int64_t tb) {
        int32_t qe[CS_ITEMS];
        bool valid[CS_ITEMS];
        const int rem = (int)((q1 - tb) < (int64_t)CS_TILE ? (q1 - tb) : (int64_t)CS_TILE);
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            qs[j] = nxt[j].x; qe[j] = nxt[j].y; qrow[j] = nxt[j].z;
            valid[j] = wv * CS_WTILE + j * kWave + lane < rem;
        }
        if (lb) {                                                              // uniform: unpack the 8-byte form {packed word, row}
            const uint32_t lmask = (1u << lb) - 1u;
#pragma unroll
            for (int j = 0; j < CS_ITEMS; ++j) {
                const uint32_t w0 = (uint32_t)qs[j];
                qrow[j] = qe[j];
                qe[j] = (int32_t)((uint32_t)smin + (w0 >> lb));
                qs[j] = (int32_t)((uint32_t)qe[j] - (w0 & lmask));
            }
        }
        if (tb + CS_TILE < q1) load_tile(tb + CS_TILE);                        // next tile's records in flight
        // hi-bound: bin of the end, then the (at most four) rows of the bin that start below it.  Rows before the bin's first
        // row start below the bin's lower edge <= end; rows of later bins start above the end: no range checks needed.
        int first[CS_ITEMS];
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            uint32_t cl = ((uint32_t)qe[j] - (uint32_t)smin) >> bshift;
            cl = cl < (uint32_t)(ncell - 1) ? cl : (uint32_t)(ncell - 1);
            first[j] = l_bin[cl];
        }
        int32_t s4[CS_ITEMS][4];
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            const int32_t* sp = l_start + first[j];
            s4[j][0] = sp[0]; s4[j][1] = sp[1]; s4[j][2] = sp[2]; s4[j][3] = sp[3];
        }
        unsigned long long more = 0;
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            unsigned long long m4;
            hi[j] = cs_count4<STRICT>(first[j], s4[j][0], s4[j][1], s4[j][2], s4[j][3], qe[j], &m4);
            more |= m4;
        }
        if (__builtin_expect(more != 0, 0)) {                                  // uniform: some probe may have more than four rows of its bin below its end
#pragma unroll
            for (int j = 0; j < CS_ITEMS; ++j) {
                if (valid[j] && hi[j] == first[j] + 4) {
                    uint32_t cl = ((uint32_t)qe[j] - (uint32_t)smin) >> bshift;
                    cl = cl < (uint32_t)(ncell - 1) ? cl : (uint32_t)(ncell - 1);
                    int lo = hi[j], hh = l_bin[cl + 1];                        // first row of [hi, rend) that does not start below the end
                    while (lo < hh) { const int mid = (lo + hh) >> 1; if (lt_op<STRICT>(l_start[mid], qe[j])) lo = mid + 1; else hh = mid; }
                    hi[j] = lo;
                }
            }
        }
        // window below hi, branch-free: sixteen ends from the 16-byte aligned row at or below hi - CS_WIN (never below row 0);
        // bit t <=> row al + t, rows at or above hi are cut off.  One prefix-max read of the row below the examined ones
        // (l_pmx[0] = the row below the slice, INT32_MIN when the contig starts here) tells whether the window runs on: those
        // probes are redone exactly.
        unsigned long long lmask = 0;
        uint32_t lngm = 0;
#pragma unroll
        for (int jj = 0; jj < CS_ITEMS; jj += 2) {
            int4 w[2][4];
            int32_t pm[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = jj + u;
                const int h = hi[j] < rk ? hi[j] : rk;
                hi[j] = h;
                int a0 = (h - CS_WIN) & ~3;
                a0 = a0 > 0 ? a0 : 0;
                al[j] = a0;
                const int4* p4 = reinterpret_cast<const int4*>(l_end + a0);
                w[u][0] = p4[0]; w[u][1] = p4[1]; w[u][2] = p4[2]; w[u][3] = p4[3];
                pm[u] = l_pmx[a0];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = jj + u;
                uint32_t m = ends_mask16<STRICT>(qs[j], w[u][0], w[u][1], w[u][2], w[u][3]);
                m = __builtin_amdgcn_ubfe(m, 0u, (uint32_t)(hi[j] - al[j]));    // rows al .. hi - 1
                const bool lng = valid[j] && lt_op<STRICT>(qs[j], pm[u]);
                mask[j] = valid[j] ? m : 0u;
                cnt[j] = __popc(mask[j]);
                lmask |= __ballot(lng);
                lngm |= lng ? (1u << j) : 0u;
            }
        }
        any_lng = lmask != 0;
        if (any_lng) {                                                         // uniform; rare: a window runs on below the examined rows
#pragma unroll
            for (int j = 0; j < CS_ITEMS; ++j) {
                if (lngm & (1u << j)) {
                    int c2 = 0;
                    walk_below(qs[j], al[j], [&](int) { ++c2; return true; });
                    cnt[j] += c2;
                }
            }
        }
    };

    // entries {wave-local probe slot << 24 | biased slice-local row} of item j into the wavefront's staging region at off;
    // returns whether one of them lies below the slice (its build row then comes from HBM at copy-out)
    auto stage_item = [&](int j, int off) -> bool {
        uint32_t m = mask[j];
        const uint32_t slot = (uint32_t)(j * kWave + lane) << 24;
        int cw = 0;
        bool below = false;
        if (any_lng) {                                                         // uniform
            cw = cnt[j] - __popc(m);                                           // matches below the window
            if (cw > 0) {
                uint32_t* sw = stw + off + cw - 1;
                walk_below(qs[j], al[j], [&](int p) { *sw-- = slot | (uint32_t)(p - r0 + CS_POS_BIAS); below = below || p < r0; return true; });
            }
        }
        const uint32_t ent = slot | (uint32_t)(al[j] + CS_POS_BIAS);
        uint32_t* so = stw + off + cw;
        while (m) {                                                            // two matches per trip: half the loop overhead
            const int t = __builtin_ctz(m);
            m &= m - 1;
            so[0] = ent + (uint32_t)t;
            if (m) { so[1] = ent + (uint32_t)__builtin_ctz(m); m &= m - 1; }
            so += 2;
        }
        return below;
    };
    // the same pairs written from the lanes (wavefronts with more pairs than the staging region holds)
    auto direct_item = [&](int j, long long off) {
        uint32_t m = mask[j];
        int cw = 0;
        if (any_lng) {
            cw = cnt[j] - __popc(m);
            if (cw > 0) {
                long long o = off + cw - 1;
                walk_below(qs[j], al[j], [&](int p) {
                    int32_t br;
                    if (p >= r0) br = l_row[p - r0];
                    else br = A.b_row[p];
                    A.out_probe[o] = qrow[j]; A.out_build[o] = br; --o;
                    return true;
                });
            }
        }
        long long o = off + cw;
        while (m) {
            const int t = __builtin_ctz(m);
            m &= m - 1;
            A.out_probe[o] = qrow[j]; A.out_build[o] = l_row[al[j] + t];
            ++o;
        }
    };
    // staged entries -> pairs, coalesced over the wavefront.  below (uniform): some entry points below the slice -- the usual
    // loop reads LDS only (a pointer select between LDS and HBM would make every read a flat load behind the stores' counter)
    auto copy_out = [&](int32_t* op, int32_t* ob, int n_ent, bool below) {
        if (!below) {
            // (the bias comes off the LDS address once, outside the loop: every entry of this loop is >= the bias)
            typedef __attribute__((address_space(3))) const int32_t lds_ci32;
            lds_ci32* rowb = (lds_ci32*)(uintptr_t)((uint32_t)(uintptr_t)(lds_ci32*)l_row - 4u * (uint32_t)CS_POS_BIAS);
            const int ca = cs_copy_align(op, A.ablate);
#pragma unroll 4
            for (int i = lane - ca; i < n_ent; i += kWave) {
                if ((unsigned)i < (unsigned)n_ent) {
                    const uint32_t e = stw[i];
                    __builtin_nontemporal_store(qrw[e >> 24], op + i);
                    __builtin_nontemporal_store((int32_t)rowb[e & 0xffffffu], ob + i);
                }
            }
            return;
        }
        for (int i = lane; i < n_ent; i += kWave) {
            const uint32_t e = stw[i];
            const int pos = (int)(e & 0xffffffu) - CS_POS_BIAS;
            int32_t br = 0;
            if (pos >= 0) br = l_row[pos];
            if (pos < 0) br = __builtin_nontemporal_load(A.b_row + (r0 + pos));
            __builtin_nontemporal_store(qrw[e >> 24], op + i);
            __builtin_nontemporal_store(br, ob + i);
        }
    };

    if constexpr (MODE != CS_FUSED) {
        // deterministic pair: slot of (workgroup v, tile tix, wavefront wv); every wavefront works on its own
        load_tile(q0);
        const int ntile = (int)((q1 - q0 + CS_TILE - 1) / CS_TILE);
        const int tiles_per_chunk = A.jchunk / CS_TILE;
        for (int tix = 0; tix < ntile; ++tix) {
            match_tile(q0 + (int64_t)tix * CS_TILE);
            const long long slot = ((long long)v * tiles_per_chunk + tix) * CS_WAVES + wv;
            int lsum = 0;
#pragma unroll
            for (int j = 0; j < CS_ITEMS; ++j) lsum += cnt[j];
            const int linc = wave_incl_sum_dpp(lsum);
            const int wtot = __builtin_amdgcn_readlane(linc, kWave - 1);
            if constexpr (MODE == CS_COUNT) {
                if (lane == 0) A.wslot[slot] = (long long)wtot;
                if (A.cache) {
                    const int64_t tb = q0 + (int64_t)tix * CS_TILE;
                    const int rem = (int)((q1 - tb) < (int64_t)CS_TILE ? (q1 - tb) : (int64_t)CS_TILE);
#pragma unroll
                    for (int j = 0; j < CS_ITEMS; ++j) {
                        const int il = wv * CS_WTILE + j * kWave + lane;
                        if (il < rem) cs_cache_store(A.cache + tb + il, cs_cache_word(al[j], mask[j], cnt[j] != __popc(mask[j])), qrow[j]);
                    }
                }
                continue;
            }
            if (wtot == 0) continue;                                           // uniform
            const long long wbase = A.wslot[slot];
            if (wtot <= A.wcap) {
                int off = linc - lsum;
                bool below = false;
#pragma unroll
                for (int j = 0; j < CS_ITEMS; ++j) {
                    qrw[j * kWave + lane] = qrow[j];
                    below = stage_item(j, off) || below;
                    off += cnt[j];
                }
                __builtin_amdgcn_wave_barrier();
                copy_out(A.out_probe + wbase, A.out_build + wbase, wtot, any_lng && __ballot(below) != 0);
                __builtin_amdgcn_wave_barrier();
            } else {
                long long off = wbase + (linc - lsum);
#pragma unroll
                for (int j = 0; j < CS_ITEMS; ++j) {
                    direct_item(j, off);
                    off += cnt[j];
                }
            }
        }
        return;
    }
    // Barrier-free tile loop (protocol of slice.hip.h's fused mode): a wavefront's pairs of a tile are contiguous in the
    // tile's output range at the offset a returning LDS atomic on the tile's cursor gives it; the LAST wavefront to arrive
    // reserves the range with the one global atomic and publishes the base in LDS; the others look at it one iteration later.
    load_tile(q0);
    const int ntile = (int)((q1 - q0 + CS_TILE - 1) / CS_TILE);
    const int spin_bound = (A.ablate & SL_ABLATE_TILE_FAULT) ? SL_SPIN_BOUND_TEST : SL_SPIN_BOUND;
    const bool tile_fault = (A.ablate & SL_ABLATE_TILE_FAULT) && v == 0;
    int pend_wtot = -1;                                                        // this wavefront's staged pairs of the previous tile
    bool pend_below = false;                                                   //   ... some of them below the slice
    long long pend_woff = 0;
    // (the last finish is peeled off the loop: see k_cs_join_plain)
    auto finish_tile = [&](int t) {
        __builtin_amdgcn_wave_barrier();
        int* c = li + (t & 1) * 4;
        unsigned long long* c64 = lc + (t & 1) * 2;
        bool timed_out = false;
        IVJ_TILE_WAIT(ld(c + 2) == 0, spin_bound, A.state, lane, timed_out = true);
        long long tb = (long long)__hip_atomic_load(c64 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (timed_out) tb = -1;
        if (tb >= 0 && pend_wtot > 0 && !(A.ablate & 32)) {
            copy_out(A.out_probe + tb + pend_woff, A.out_build + tb + pend_woff, pend_wtot, pend_below);
        }
        if (lane == 0) {
            if (atomicAdd(c + 1, 1) == CS_WAVES - 1) {                         // last wavefront out: recycle the block for tile t + 2
                __hip_atomic_store(c64, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                stv(c + 0, 0); stv(c + 1, 0); stv(c + 2, 0);
                stv(c + 3, t + 2);
            }
        }
    };
    for (int tix = 0; tix < ntile; ++tix) {
        match_tile(q0 + (int64_t)tix * CS_TILE);
        if (tix > 0) finish_tile(tix - 1);                                     // its base was requested one iteration ago
        int* c = li + (tix & 1) * 4;
        unsigned long long* c64 = lc + (tix & 1) * 2;
        IVJ_TILE_WAIT(ld(c + 3) != tix, spin_bound, A.state, lane, return);       // the block is ours (recycled after tile tix - 2)
        int lsum = 0;
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) lsum += cnt[j];
        const int linc = wave_incl_sum_dpp(lsum);
        const int wtot = __builtin_amdgcn_readlane(linc, kWave - 1);
        long long woff = 0;
        if (lane == 0) {
            woff = (long long)atomicAdd(c64, (unsigned long long)wtot);
            if (atomicAdd(c + 0, 1) == CS_WAVES - 1) {                         // last wavefront in: the tile's total is complete
                const long long total = (long long)__hip_atomic_load(c64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                long long base = 0;
                if (total > 0) {
                    base = (long long)atomicAdd(&A.state[0], (unsigned long long)total);
                    if (base + total > A.capacity) { atomicOr(&A.state[1], 1ull); base = -1; }
                }
                __hip_atomic_store(c64 + 1, (unsigned long long)base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!(tile_fault && tix == 0)) stv(c + 2, 1);
            }
        }
        woff = ((long long)__builtin_amdgcn_readfirstlane((int)(woff >> 32)) << 32) | (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(woff & 0xffffffffll));
        if (wtot > 0 && wtot <= A.wcap) {
            // usual case: the wavefront's entries into its staging region, copied out one iteration later
            int off = linc - lsum;
            bool below = false;
#pragma unroll
            for (int j = 0; j < CS_ITEMS; ++j) {
                qrw[j * kWave + lane] = qrow[j];
                below = stage_item(j, off) || below;
                off += cnt[j];
            }
            pend_wtot = wtot;
            pend_below = any_lng && __ballot(below) != 0;
        } else if (wtot > 0) {
            // dense wavefront: wait for the base now, write from the lanes
            bool timed_out = false;
            IVJ_TILE_WAIT(ld(c + 2) == 0, spin_bound, A.state, lane, timed_out = true);
            long long tb = (long long)__hip_atomic_load(c64 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (timed_out) tb = -1;
            if (tb >= 0) {
                long long off = tb + woff + (linc - lsum);
#pragma unroll
                for (int j = 0; j < CS_ITEMS; ++j) {
                    direct_item(j, off);
                    off += cnt[j];
                }
            }
            pend_wtot = 0;
        } else pend_wtot = 0;
        pend_woff = woff;
    }
    if (ntile > 0) finish_tile(ntile - 1);
    cs_publish_state(A, total_wg);
}


// ---- FILL from the cache: the second pass of the count -> fill pair without a second matching pass ---------------------------------
// COUNT left {word, probe row} per probe record (cs_cache_word) and one scanned slot per (tile, wavefront).  This kernel streams
// those 8 bytes per probe (not the 12-byte records), turns mask bits into pairs and writes them at the wavefront's base: no bins, no
// starts, no ends, no window compare -- the slice's build ROWS are the only index data in LDS (4 bytes per row), so two workgroups
// share a CU.  Same workgroup -> (bucket, chunk) map, same tiles, same wavefront shares as COUNT (the slots must line up).
// Flagged words (a window running on below the examined rows) redo exactly what COUNT did for them, reading the ends and prefix
// maxima from HBM: WALK = false recounts row by row from the hi-bound (k_cs_join_plain's rule), WALK = true walks the block maxima
// below al (k_cs_join's rule).
struct CsFillLds { int row, qrow, stage, total; };
__host__ __device__ inline CsFillLds cs_fill_lds(int R, int wcap) {
    CsFillLds L;
    int o = 0;
    L.row = o; o += 4 * R;
    L.qrow = (o + 15) & ~15; o = L.qrow + 4 * CS_TILE;
    L.stage = o; o += 4 * wcap * CS_WAVES;
    L.total = o;
    return L;
}

template <bool STRICT, bool WALK, bool TWO>
__global__ __launch_bounds__(CS_THREADS, TWO ? 8 : 4) void k_cs_fill(CsJoinArgs A) {      // (second bound: wavefronts per SIMD -- 8 = two workgroups per CU)
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
    const CsFillLds L = cs_fill_lds(A.R, A.wcap);
    int32_t* l_row = reinterpret_cast<int32_t*>(cs_lds + L.row);
    int32_t* l_qrow = reinterpret_cast<int32_t*>(cs_lds + L.qrow);
    uint32_t* l_stage = reinterpret_cast<uint32_t*>(cs_lds + L.stage);

    const int total_wg = A.meta[0];
    const int per = (total_wg + 7) / 8;
    const int wslot = (int)(blockIdx.x >> 3);
    const int v = (int)(blockIdx.x & 7) * per + wslot;
    if (wslot >= per || v >= total_wg) return;                                 // uniform
    const int2 bc = A.wg_map[v];
    const int k = bc.x;
    const int64_t q0 = (int64_t)A.bstart[k] + (int64_t)bc.y * A.jchunk;
    const int64_t qend = A.bend ? (int64_t)A.bend[k] : (int64_t)A.bstart[k + 1];
    const int64_t q1 = q0 + A.jchunk < qend ? q0 + A.jchunk : qend;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
    const int4 sm1 = A.smeta[2 * k + 1];
    const int seg_a = sm1.x, rk = sm1.z, r0 = sm1.w;
    const int lb = A.meta[CS_META_FMT];                                        // record format of the call (uniform)
    const int32_t smin = A.smeta[2 * k].x;
    for (int i = tid; i < rk; i += CS_THREADS) l_row[i] = A.b_row[r0 + i];
    __syncthreads();

    uint32_t* stw = l_stage + wv * A.wcap;
    int32_t* qrw = l_qrow + wv * CS_WTILE;
    auto ep_at = [&](int p) -> int2 { return A.ep[p]; };
    // k_cs_join's walk over the rows below slice-local row a0, every row read from HBM
    auto walk_below = [&](int32_t qsv, int a0, auto&& f) {
        int i = a0 - 1;
        const int stop = a0 - CS_LIN > 0 ? a0 - CS_LIN : 0;
        for (; i >= stop; --i) {
            const int2 e = A.ep[r0 + i];
            if (!lt_op<STRICT>(qsv, e.y)) return;
            if (lt_op<STRICT>(qsv, e.x)) { if (!f(r0 + i)) return; }
        }
        if (r0 + i >= seg_a) hier_walk<STRICT>(A.hier, ep_at, seg_a, r0 + i, qsv, f);
    };

    const int ntile = (int)((q1 - q0 + CS_TILE - 1) / CS_TILE);
    const int tiles_per_chunk = A.jchunk / CS_TILE;
    // records and words of the next tile are requested before the current one is emitted
    int32_t n_row[CS_ITEMS];
    uint32_t n_w[CS_ITEMS];
    auto load_tile = [&](int64_t tb) {
        const int rem = (int)((q1 - tb) < (int64_t)CS_TILE ? (q1 - tb) : (int64_t)CS_TILE);
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            const int il = wv * CS_WTILE + j * kWave + lane;
            n_row[j] = -1; n_w[j] = 0u;
            if (il < rem) {
                const uint2 c = cs_cache_load(A.cache + tb + il);
                n_w[j] = c.x; n_row[j] = (int32_t)c.y;
            }
        }
    };
    if (!TWO && ntile > 0) load_tile(q0);
    for (int tix = 0; tix < ntile; ++tix) {
        int32_t qs[CS_ITEMS], qrow[CS_ITEMS];
        uint32_t w[CS_ITEMS];
        const int64_t tb0 = q0 + (int64_t)tix * CS_TILE;
        if (TWO) load_tile(tb0);                                               // two workgroups per CU: the other one covers the latency, no prefetch registers
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            qrow[j] = n_row[j]; w[j] = n_w[j];
            qs[j] = 0;
            // (rare) the probe's start is only needed to redo a running-on window: fetched here, not prefetched
            if (w[j] & CS_CACHE_FLAG) {
                const int64_t ri = tb0 + wv * CS_WTILE + j * kWave + lane;
                if (lb) { const uint32_t w0 = (uint32_t)A.rec[2 * ri]; qs[j] = (int32_t)((uint32_t)smin + (w0 >> lb) - (w0 & ((1u << lb) - 1u))); }
                else qs[j] = A.rec[3 * ri];
            }
        }
        if (!TWO && tix + 1 < ntile) load_tile(q0 + (int64_t)(tix + 1) * CS_TILE);
        int al[CS_ITEMS], cnt[CS_ITEMS];
        uint32_t mask[CS_ITEMS];
        uint32_t flg = 0;
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) {
            al[j] = (int)(w[j] & 0x1fffu);
            mask[j] = (w[j] >> 13) & 0xffffu;
            cnt[j] = __popc(mask[j]);
            flg |= (w[j] & CS_CACHE_FLAG) ? (1u << j) : 0u;
        }
        const bool any_flag = __ballot(flg != 0) != 0;                         // uniform
        if (any_flag) {
#pragma unroll
            for (int j = 0; j < CS_ITEMS; ++j) {
                if (flg & (1u << j)) {
                    int c2 = 0;
                    if constexpr (WALK) { walk_below(qs[j], al[j], [&](int) { ++c2; return true; }); cnt[j] += c2; }
                    else {
                        for (int p = r0 + al[j] - 1; p >= seg_a; --p) {        // al = the hi-bound here: every row below it, as COUNT walked them
                            const int2 e = A.ep[p];
                            if (!lt_op<STRICT>(qs[j], e.y)) break;
                            c2 += lt_op<STRICT>(qs[j], e.x) ? 1 : 0;
                        }
                        cnt[j] = c2;
                    }
                }
            }
        }
        int lsum = 0;
#pragma unroll
        for (int j = 0; j < CS_ITEMS; ++j) lsum += cnt[j];
        const int linc = wave_incl_sum_dpp(lsum);
        const int wtot = __builtin_amdgcn_readlane(linc, kWave - 1);
        if (wtot == 0) continue;                                               // uniform
        const long long slot = ((long long)v * tiles_per_chunk + tix) * CS_WAVES + wv;
        const long long wbase = A.wslot[slot];
        if (wtot <= A.wcap && (WALK || !any_flag)) {
            int off = linc - lsum;
            if constexpr (!WALK) {
                // the usual case: entries {wave-local probe slot << 16 | slice-local row}, resolved at copy-out
#pragma unroll
                for (int j = 0; j < CS_ITEMS; ++j) {
                    qrw[j * kWave + lane] = qrow[j];
                    uint32_t m = mask[j];
                    const uint32_t ent = ((uint32_t)(j * kWave + lane) << 16) | (uint32_t)al[j];
                    uint32_t* so = stw + off;
                    while (m) {
                        const int t = __builtin_ctz(m);
                        m &= m - 1;
                        so[0] = ent + (uint32_t)t;
                        if (m) { so[1] = ent + (uint32_t)__builtin_ctz(m); m &= m - 1; }
                        so += 2;
                    }
                    off += cnt[j];
                }
                __builtin_amdgcn_wave_barrier();
                int32_t* op = A.out_probe + wbase;
                int32_t* ob = A.out_build + wbase;
                if (TWO || (A.ablate & CS_ABLATE_STORE4)) {                     // (four bytes per lane and store: rounds 3-5, A/B runs; the 64-register form of two workgroups per CU)
                    const int ca = cs_copy_align(op, A.ablate);
#pragma unroll 4
                    for (int i = lane - ca; i < wtot; i += kWave) {
                        if ((unsigned)i < (unsigned)wtot) {
                            const uint32_t e = stw[i];
                            __builtin_nontemporal_store(qrw[e >> 16], op + i);
                            __builtin_nontemporal_store(l_row[e & 0xffffu], ob + i);
                        }
                    }
                } else cs_copy_pairs16<16>(stw, qrw, l_row, op, ob, wtot, lane);
                __builtin_amdgcn_wave_barrier();
            } else {
                // k_cs_join's entries {wave-local probe slot << 24 | (row - first row of the slice) + bias}: rows below the slice
                // (reached by the walk) are staged too and take their build row from HBM at copy-out
                bool below = false;
#pragma unroll
                for (int j = 0; j < CS_ITEMS; ++j) {
                    qrw[j * kWave + lane] = qrow[j];
                    uint32_t m = mask[j];
                    const uint32_t pslot = (uint32_t)(j * kWave + lane) << 24;
                    int cw = 0;
                    if (any_flag && ((flg >> j) & 1u)) {
                        cw = cnt[j] - __popc(m);
                        if (cw > 0) {
                            uint32_t* sw = stw + off + cw - 1;
                            walk_below(qs[j], al[j], [&](int p) { *sw-- = pslot | (uint32_t)(p - r0 + CS_POS_BIAS); below = below || p < r0; return true; });
                        }
                    }
                    const uint32_t ent = pslot | (uint32_t)(al[j] + CS_POS_BIAS);
                    uint32_t* so = stw + off + cw;
                    while (m) {
                        const int t = __builtin_ctz(m);
                        m &= m - 1;
                        so[0] = ent + (uint32_t)t;
                        if (m) { so[1] = ent + (uint32_t)__builtin_ctz(m); m &= m - 1; }
                        so += 2;
                    }
                    off += cnt[j];
                }
                const bool any_below = any_flag && __ballot(below) != 0;
                __builtin_amdgcn_wave_barrier();
                int32_t* op = A.out_probe + wbase;
                int32_t* ob = A.out_build + wbase;
                if (!any_below && !(A.ablate & CS_ABLATE_STORE4)) {            // (uniform) every row in LDS: sixteen bytes per lane and store
                    typedef __attribute__((address_space(3))) const int32_t lds_ci32;
                    lds_ci32* rowb = (lds_ci32*)(uintptr_t)((uint32_t)(uintptr_t)(lds_ci32*)l_row - 4u * (uint32_t)CS_POS_BIAS);
                    cs_copy_pairs16<24>(stw, qrw, rowb, op, ob, wtot, lane);
                } else {
                    for (int i = lane; i < wtot; i += kWave) {
                        const uint32_t e = stw[i];
                        const int pos = (int)(e & 0xffffffu) - CS_POS_BIAS;
                        int32_t br = 0;
                        if (!any_below || pos >= 0) br = l_row[pos];
                        if (any_below && pos < 0) br = __builtin_nontemporal_load(A.b_row + (r0 + pos));
                        __builtin_nontemporal_store(qrw[e >> 24], op + i);
                        __builtin_nontemporal_store(br, ob + i);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            // a wavefront with more pairs than its staging holds, or with a window that ran on: written from the lanes
            long long off = wbase + (linc - lsum);
#pragma unroll
            for (int j = 0; j < CS_ITEMS; ++j) {
                const bool f = (flg >> j) & 1u;
                if (WALK || !f) {
                    int cw = 0;
                    if (WALK && f) {
                        cw = cnt[j] - __popc(mask[j]);
                        long long o = off + cw - 1;                            // matches below the window first in the range, in ascending position
                        walk_below(qs[j], al[j], [&](int p) {
                            A.out_probe[o] = qrow[j]; A.out_build[o] = p >= r0 ? l_row[p - r0] : A.b_row[p]; --o;
                            return true;
                        });
                    }
                    uint32_t m = mask[j];
                    long long o = off + cw;
                    while (m) {
                        const int t = __builtin_ctz(m);
                        m &= m - 1;
                        A.out_probe[o] = qrow[j]; A.out_build[o] = l_row[al[j] + t];
                        ++o;
                    }
                } else {
                    long long o = off + cnt[j] - 1;                            // the f-th match from the top owns slot end - 1 - f
                    for (int p = r0 + al[j] - 1; o >= off; --p) {
                        if (lt_op<STRICT>(qs[j], A.ep[p].x)) {
                            A.out_probe[o] = qrow[j]; A.out_build[o] = p >= r0 ? l_row[p - r0] : A.b_row[p]; --o;
                        }
                    }
                }
                off += cnt[j];
            }
        }
    }
}

}  // namespace ivj
