// index_build.hip.h -- kernels that turn the radix-sorted build side into the index of index_view.hip.h
// (sorted columns, segment offsets, prefix max, direct-address tables and their 16-byte records,
// end order, nearest candidate records).
#pragma once
#include "index_view.hip.h"

namespace ivj {

// ------------------------------------------------------------------ index build

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
// fall into one bin: ONE 32-byte record (two 16-byte reads of one line) answers both ranks, with the keys
// of the bin's first three rows of either order inline (no dependent key gather unless the bin is crowded).
__global__ void k_contig_meta_joint(const int32_t* __restrict__ seg, const int32_t* __restrict__ b_start,
                                    const int32_t* __restrict__ e_end, int32_t n_contigs, int bins_per_row,
                                    int4* __restrict__ cmeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_contigs) return;
    const int a = seg[c], b = seg[c + 1];
    uint32_t ulo = 0, uhi = 0;
    int shift = 0;
    if (b > a) {
        const uint32_t s0 = flip(b_start[a]), e0 = flip(e_end[a]), s1 = flip(b_start[b - 1]), e1 = flip(e_end[b - 1]);
        ulo = s0 < e0 ? s0 : e0; uhi = s1 > e1 ? s1 : e1;
        unsigned long long span = (unsigned long long)(uhi - ulo), cap = (unsigned long long)bins_per_row * (unsigned long long)(b - a);
        if (cap < 2) cap = 2;                                  // the table has 2 (b - a) + 2 slots; keeps shift <= 31
        while ((span >> shift) + 1ull > cap) ++shift;
    }
    cmeta[2 * c] = make_int4(a, b, (int)ulo, (int)uhi);
    cmeta[2 * c + 1] = make_int4(shift, 2 * a + 2 * c, 0, 0);
}

// Joint-grid record of count_overlaps, 16 bytes per bin, so that BOTH ranks of a probe come out of ONE gather:
//   {first start position | more << 31, first end position | more << 31, start offsets o0 | o1 << 16, end offsets o0 | o1 << 16}
// o0, o1 = the first two keys of the bin as 16-bit offsets from the bin's lower edge (0xffff: no such row -- never below a
// target, whose own offset is < 2^shift <= 0xffff); "more" = the bin holds a third row of that order, which the two
// offsets cannot answer.  Bins wider than 2^16 (a build side of fewer than ~span / 2^17 rows per contig) carry no
// offsets: "more" then means "the bin is not empty" and the rank is searched from the bin's first position.
// bins_s / bins_e are the max-scanned first positions, so the difference of two neighbouring slots is the number of rows
// in the bin (the slot after a contig's last bin holds the segment end).
__global__ void k_joint_records(const uint32_t* __restrict__ bins_s, const uint32_t* __restrict__ bins_e, int64_t bins_len,
                                const int32_t* __restrict__ b_start, const int32_t* __restrict__ e_end,
                                const int4* __restrict__ cmeta, int32_t n_contigs, int4* __restrict__ jrec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bins_len) return;
    int lo = 0, hi = n_contigs;
    while (lo < hi) { const int m = (lo + hi) >> 1; if ((int64_t)cmeta[2 * m + 1].y <= i) lo = m + 1; else hi = m; }
    const int c = lo - 1;
    const uint32_t ps = bins_s[i], pe = bins_e[i];
    uint32_t ws = ps, we = pe, os = 0xffffffffu, oe = 0xffffffffu;
    if (c >= 0) {
        const int4 m0 = cmeta[2 * c], m1 = cmeta[2 * c + 1];
        const uint32_t bend = (uint32_t)m0.y;
        const int shift = m1.x;
        const uint32_t edge = (uint32_t)m0.z + (uint32_t)(((unsigned long long)(i - (int64_t)m1.y)) << shift);   // never used past the contig's last bin
        uint32_t ns = i + 1 < bins_len ? bins_s[i + 1] : bend, ne = i + 1 < bins_len ? bins_e[i + 1] : bend;
        ns = ns < bend ? ns : bend; ne = ne < bend ? ne : bend;
        const uint32_t cs = ns > ps ? ns - ps : 0u, ce = ne > pe ? ne - pe : 0u;
        if (shift <= 16) {
            const uint32_t s0 = cs >= 1 ? flip(b_start[ps]) - edge : 0xffffu, s1 = cs >= 2 ? flip(b_start[ps + 1]) - edge : 0xffffu;
            const uint32_t e0 = ce >= 1 ? flip(e_end[pe]) - edge : 0xffffu, e1 = ce >= 2 ? flip(e_end[pe + 1]) - edge : 0xffffu;
            os = (s0 & 0xffffu) | (s1 << 16); oe = (e0 & 0xffffu) | (e1 << 16);
            if (cs >= 3) ws |= 0x80000000u;
            if (ce >= 3) we |= 0x80000000u;
        } else {
            if (cs >= 1) ws |= 0x80000000u;
            if (ce >= 1) we |= 0x80000000u;
        }
    }
    jrec[i] = make_int4((int)ws, (int)we, (int)os, (int)oe);
}

// nearest (k = 1): everything a probe needs about its bound position p in ONE 32-byte record (two 16-byte halves of one
// line, requested together):
//   nrec[2p]     = {pmax[p-1], build row of the first row attaining it, start[p], end[p]}   left / right candidate
//   nrec[2p + 1] = {build row of row p (-1: p = n), v1, row1, v2}                            earlier prefix-max levels
// The prefix max below p is a staircase of levels (positions where it rises: pargmax); for an overlapping probe the
// answer is the first row of the EARLIEST level whose value is still above q.start.  Level m (the current one) is in the
// first half; v1 / row1 = value and first row of level m-1 (row1 = -1: no such level inside the contig segment), v2 =
// value of level m-2 (INT32_MIN: none) -- a probe below v2 too falls back to the bound search, which is always right.
// n + 1 records; the fields that do not exist (p = 0 / p = n) are never read.
__global__ void k_nearest_records(const int32_t* __restrict__ b_start, const int2* __restrict__ ep,
                                  const int32_t* __restrict__ b_row, const int32_t* __restrict__ b_contig,
                                  const int32_t* __restrict__ pargmax, int64_t n, int4* __restrict__ nrec) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p > n) return;
    int4 r = make_int4(0, -1, 0, 0), q = make_int4(-1, 0, -1, (int)0x80000000);
    if (p >= 1) {
        const int rm = pargmax[p - 1];
        r.x = ep[p - 1].y; r.y = b_row[rm];
        if (rm >= 1 && b_contig[rm - 1] == b_contig[rm]) {
            const int r1 = pargmax[rm - 1];
            q.y = ep[rm - 1].y; q.z = b_row[r1];
            if (r1 >= 1 && b_contig[r1 - 1] == b_contig[r1]) q.w = ep[r1 - 1].y;
        }
    }
    if (p < n) { r.z = b_start[p]; r.w = ep[p].x; q.x = b_row[p]; }
    nrec[2 * p] = r;
    nrec[2 * p + 1] = q;
}

// Nearest lines (round 5: 64 bytes per table slot, 128 bytes per build row; round 4: 128-byte lines).  A probe whose end falls into
// slot i has its hi-bound in {p0 .. p0 + rows of the bin}, p0 = the slot's first position, and with two bins per build row 98.6 % of the
// bins hold at most two rows -- so the line carries what the probe needs for hi = p0, p0 + 1, p0 + 2.  Round 4 stored the three 32-byte
// nearest records side by side.  But the record of position p + 1 FOLLOWS from the record of p and row p: the prefix max either stays
// (end[p] <= pmax[p-1]: same levels) or row p opens a new level on top (the old levels move one down), so the line holds ONE record and
// three rows and the kernel replays at most two "pushes" in registers:
//   word 0       p0 | first << 30          first: p0 is the first row of its contig's segment (nothing below it)
//   words 1-5    pmax[p0-1], its first row, value and first row of the level below, value of the level below that   (= nrec of p0)
//   words 6-14   {start, end, build row} of the rows p0, p0 + 1, p0 + 2   (rows past the segment: start INT32_MAX, never below a target)
//   word 15      spare
// The rank of the probe inside the bin is the number of the three starts below its end (rows of later bins start above it); three
// below: a fourth row may follow -- left to the two-gather kernel like every probe a line cannot settle.  One line fetch of 64 bytes per
// probe (gather_probe: the price of a gather from beyond the L2 is per REQUEST, ~ 20 ps, whatever its width up to a line -- but the
// round-4 line was two 64-byte requests).  Four threads per slot, 16 bytes each: coalesced non-temporal writes.
__global__ void k_nearest_lines(const uint32_t* __restrict__ bins, const int4* __restrict__ cmeta, int32_t n_contigs, const int4* __restrict__ nrec,
                                const int32_t* __restrict__ b_start, const int2* __restrict__ ep, const int32_t* __restrict__ b_row,
                                int64_t slots, int64_t n, int4* __restrict__ nline) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t slot = t >> 2;
    const int w = (int)(t & 3);
    // last contig whose table offset tb = cmeta[2c+1].y is <= slot: ONE bound search per workgroup (its first slot), then a step forward
    // where a contig's table ends inside the workgroup's 64 slots (round 6; the per-thread search was a chain of ~ 5 dependent loads in
    // front of the three dependent levels the line itself needs)
    __shared__ int s_c;
    if (threadIdx.x == 0) {
        const int64_t first = ((int64_t)blockIdx.x * blockDim.x) >> 2;
        int lo = 0, hi = n_contigs;
        while (lo < hi) { const int m = (lo + hi) >> 1; if ((int64_t)cmeta[2 * m + 1].y <= first) lo = m + 1; else hi = m; }
        s_c = lo - 1;
    }
    __syncthreads();
    if (slot >= slots) return;
    int c = s_c;
    while (c + 1 < n_contigs && (int64_t)cmeta[2 * (c + 1) + 1].y <= slot) ++c;
    int4 v = make_int4(0, 0, 0, 0);
    if (c >= 0) {
        const int4 m0 = cmeta[2 * c];
        const int64_t a = m0.x, b = m0.y;
        int64_t p0 = (int64_t)bins[slot];
        p0 = p0 < b ? p0 : b;
        auto row = [&](int j, int32_t& s, int32_t& e, int32_t& r) {
            const int64_t p = p0 + j;
            s = INT32_MAX; e = 0; r = -1;
            if (p < b) { s = b_start[p]; e = ep[p].x; r = b_row[p]; }
        };
        const int4 R = nrec[2 * p0], Q = nrec[2 * p0 + 1];
        int32_t s0, e0, r0, s1, e1, r1, s2, e2, r2;
        if (w == 0) v = make_int4((int)((uint32_t)p0 | (p0 == a ? 0x40000000u : 0u)), R.x, R.y, Q.y);
        else if (w == 1) { row(0, s0, e0, r0); v = make_int4(Q.z, Q.w, s0, e0); }
        else if (w == 2) { row(0, s0, e0, r0); row(1, s1, e1, r1); v = make_int4(r0, s1, e1, r1); }
        else { row(2, s2, e2, r2); v = make_int4(s2, e2, r2, 0); }
    }
    // one 16-byte store per thread (written as four component stores until round 6; the compiler had merged them: 0.089 ms either way)
    typedef int nl_v4 __attribute__((ext_vector_type(4)));
    nl_v4 vv; vv.x = v.x; vv.y = v.y; vv.z = v.z; vv.w = v.w;
    __builtin_nontemporal_store(vv, reinterpret_cast<nl_v4*>(nline + t));
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

// brec[i] for table slot i (p0 = bins[i] after the max-scan; bins[i + 1] - bins[i] = rows in the bin; the slot after a
// contig's last bin holds the segment end).  Bins at most 2^16 wide: {p0 | more << 31, o0 | o1 << 16, o2 | o3 << 16,
// o4 | o5 << 16}, o_j = key of row p0 + j as an offset from the bin's lower edge, 0xffff when the bin has no such row
// (a target's own offset is < 2^shift <= 0xffff + 1, so 0xffff never counts as below it); more = a seventh row exists.
// Wider bins: {p0 | more << 31, key[p0], key[p0+1], key[p0+2]} (keys past the contig segment: INT32_MAX), more = the bin
// holds a fourth row.  The contig of a slot is found by a bound search over the table offsets.
__global__ void k_bins_records(const uint32_t* __restrict__ bins, int64_t bins_len, const int32_t* __restrict__ keys,
                               const int4* __restrict__ cmeta, int32_t n_contigs, int4* __restrict__ brec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bins_len) return;
    // last contig whose table offset tb = cmeta[2c+1].y is <= i
    int lo = 0, hi = n_contigs;
    while (lo < hi) { const int m = (lo + hi) >> 1; if ((int64_t)cmeta[2 * m + 1].y <= i) lo = m + 1; else hi = m; }
    const int c = lo - 1;
    const uint32_t p0 = bins[i];
    int4 r = make_int4((int)p0, 0x7fffffff, 0x7fffffff, 0x7fffffff);
    if (c >= 0) {
        const int4 m0 = cmeta[2 * c], m1 = cmeta[2 * c + 1];
        const uint32_t bend = (uint32_t)m0.y;
        uint32_t nx = i + 1 < bins_len ? bins[i + 1] : bend;
        nx = nx < bend ? nx : bend;
        const uint32_t cnt = nx > p0 ? nx - p0 : 0u;
        if (m1.x <= 16) {
            const uint32_t edge = (uint32_t)m0.z + (uint32_t)(((unsigned long long)(i - (int64_t)m1.y)) << m1.x);
            uint32_t o[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) o[j] = (uint32_t)j < cnt ? ((flip(keys[p0 + j]) - edge) & 0xffffu) : 0xffffu;
            r = make_int4((int)(p0 | (cnt > 6u ? 0x80000000u : 0u)), (int)(o[0] | (o[1] << 16)), (int)(o[2] | (o[3] << 16)), (int)(o[4] | (o[5] << 16)));
        } else {
            if (p0 < bend) r.y = keys[p0];
            if (p0 + 1 < bend) r.z = keys[p0 + 1];
            if (p0 + 2 < bend) r.w = keys[p0 + 2];
            r.x = (int)(p0 | (cnt > 3u ? 0x80000000u : 0u));
        }
    }
    brec[i] = r;
}

}  // namespace ivj
