/*
 * ivj_oracle.c -- CPU restatement of the interval-join hot path.
 * TEST INFRASTRUCTURE ONLY (see ivj_oracle.h for the reference citations and
 * the import rule).  Plain C11 + OpenMP; integer arithmetic only.
 */
#include "ivj_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- the predicate: polars_bio/range_op.py:75-84 ------------------------
 * Strict (0-based half-open): a.start <  b.end && b.start <  a.end
 * Weak   (1-based closed)   : a.start <= b.end && b.start <= a.end        */
static inline int lt_op(int32_t x, int32_t y, int strict) { return strict ? (x < y) : (x <= y); }

static inline int cond_a(int32_t bs, int32_t qe, int strict) { return lt_op(bs, qe, strict); } /* b.start (<) q.end */
static inline int cond_b(int32_t qs, int32_t be, int strict) { return lt_op(qs, be, strict); } /* q.start (<) b.end */

static inline int64_t gap_dist(int32_t qs, int32_t qe, int32_t bs, int32_t be) {
    int64_t d1 = (int64_t)bs - (int64_t)qe;
    int64_t d2 = (int64_t)qs - (int64_t)be;
    return d1 > d2 ? d1 : d2;
}

/* ======================= brute force ===================================== */

typedef struct { int32_t start; int32_t row; } sr_t;
static int cmp_sr(const void* a, const void* b) {
    const sr_t* x = (const sr_t*)a; const sr_t* y = (const sr_t*)b;
    if (x->start != y->start) return x->start < y->start ? -1 : 1;
    return x->row < y->row ? -1 : (x->row > y->row);
}

int64_t orc_overlap_brute(const orc_side* probe, const orc_side* build, int strict,
                          int32_t* out_probe, int32_t* out_build, int64_t cap) {
    int64_t total = 0;
    sr_t* hits = (sr_t*)malloc(sizeof(sr_t) * (size_t)(build->n > 0 ? build->n : 1));
    for (int64_t i = 0; i < probe->n; ++i) {
        int64_t nh = 0;
        for (int64_t j = 0; j < build->n; ++j) {
            if (probe->contig[i] != build->contig[j]) continue;
            if (cond_b(probe->start[i], build->end[j], strict) &&
                cond_a(build->start[j], probe->end[i], strict)) {
                hits[nh].start = build->start[j]; hits[nh].row = (int32_t)j; ++nh;
            }
        }
        qsort(hits, (size_t)nh, sizeof(sr_t), cmp_sr);
        for (int64_t h = 0; h < nh; ++h) {
            if (out_probe && total < cap) { out_probe[total] = (int32_t)i; out_build[total] = hits[h].row; }
            ++total;
        }
    }
    free(hits);
    return total;
}

void orc_count_overlaps_brute(const orc_side* probe, const orc_side* build, int strict,
                              int64_t* counts) {
    for (int64_t i = 0; i < probe->n; ++i) {
        int64_t c = 0;
        for (int64_t j = 0; j < build->n; ++j) {
            if (probe->contig[i] != build->contig[j]) continue;
            c += cond_b(probe->start[i], build->end[j], strict) &&
                 cond_a(build->start[j], probe->end[i], strict);
        }
        counts[i] = c;
    }
}

typedef struct { int64_t d; int32_t cls; int32_t start; int32_t row; } cand_t;
static int cmp_cand(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->d != y->d) return x->d < y->d ? -1 : 1;
    if (x->cls != y->cls) return x->cls < y->cls ? -1 : 1;
    if (x->start != y->start) return x->start < y->start ? -1 : 1;
    return x->row < y->row ? -1 : (x->row > y->row);
}

void orc_nearest_brute(const orc_side* probe, const orc_side* build, int strict,
                       int k, int include_overlaps,
                       int32_t* out_idx, int64_t* out_dist, int32_t* out_n) {
    cand_t* c = (cand_t*)malloc(sizeof(cand_t) * (size_t)(build->n > 0 ? build->n : 1));
    for (int64_t i = 0; i < probe->n; ++i) {
        int64_t nc = 0;
        int32_t qs = probe->start[i], qe = probe->end[i];
        for (int64_t j = 0; j < build->n; ++j) {
            if (probe->contig[i] != build->contig[j]) continue;
            int a = cond_a(build->start[j], qe, strict);
            int b = cond_b(qs, build->end[j], strict);
            cand_t t;
            if (a && b) { if (!include_overlaps) continue; t.d = 0; t.cls = 0; }
            else { t.d = gap_dist(qs, qe, build->start[j], build->end[j]); t.cls = a ? 1 : 2; }
            t.start = build->start[j]; t.row = (int32_t)j;
            c[nc++] = t;
        }
        qsort(c, (size_t)nc, sizeof(cand_t), cmp_cand);
        int32_t n = (int32_t)(nc < k ? nc : k);
        for (int32_t r = 0; r < k; ++r) {
            out_idx[i * k + r] = r < n ? c[r].row : -1;
            out_dist[i * k + r] = r < n ? c[r].d : -1;
        }
        out_n[i] = n;
    }
    free(c);
}

/* ======================= sort + bound search ============================= */

struct orc_index {
    int64_t n;
    int n_contigs;
    int has_inverted;     /* any build row with start > end */
    int64_t* seg;         /* n_contigs + 1 offsets into the sorted arrays */
    int32_t* s_start;     /* sorted by (contig, start, row) */
    int32_t* s_end;       /* end in that order */
    int32_t* s_row;       /* original build row */
    int32_t* pmax;        /* prefix max of s_end inside the contig segment */
    int32_t* e_end;       /* ends sorted by (contig, end, position in start order) */
    int32_t* e_pos;       /* position in start order of that end */
    int32_t* tmax;        /* implicit interval tree: max end in the subtree whose root is this position */
};

/* LSD radix sort of 64-bit keys with a 32-bit payload, 16-bit digits.  Large inputs: every pass is cut into one contiguous
 * chunk per thread (per-thread digit histograms -> offsets by (digit, thread) -> stable scatter), so that the all-core CPU
 * baseline of bench.py is not charged a single-threaded index build. */
static void radix_sort_u64(uint64_t* keys, int32_t* vals, int64_t n, int key_bits) {
    if (n <= 1) return;
    uint64_t* k2 = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
    int32_t* v2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    uint64_t* src = keys; uint64_t* dst = k2; int32_t* vs = vals; int32_t* vd = v2;
    int nt = 1;
#ifdef _OPENMP
    /* (inside a parallel region -- the per-thread sorts of orc_overlap_baseline -- a nested team would have one thread) */
    if (n >= (1 << 20) && !omp_in_parallel()) { nt = omp_get_max_threads(); if (nt > 64) nt = 64; if (nt < 1) nt = 1; }
#endif
    int64_t* cnt = (int64_t*)malloc(sizeof(int64_t) * 65536 * (size_t)nt);
    for (int shift = 0; shift < key_bits; shift += 16) {
        memset(cnt, 0, sizeof(int64_t) * 65536 * (size_t)nt);
        /* (a loop over the nt chunks, not "one chunk per thread of the team": the runtime may hand out fewer threads than asked for) */
#pragma omp parallel for schedule(static, 1) num_threads(nt)
        for (int t = 0; t < nt; ++t) {
            const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
            int64_t* c = cnt + (size_t)t * 65536;
            for (int64_t i = lo; i < hi; ++i) c[(src[i] >> shift) & 0xFFFF]++;
        }
        int64_t run = 0;
        for (int d = 0; d < 65536; ++d)
            for (int t = 0; t < nt; ++t) { int64_t* c = cnt + (size_t)t * 65536 + d; const int64_t v = *c; *c = run; run += v; }
#pragma omp parallel for schedule(static, 1) num_threads(nt)
        for (int t = 0; t < nt; ++t) {
            const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
            int64_t* c = cnt + (size_t)t * 65536;
            for (int64_t i = lo; i < hi; ++i) {
                const int64_t p = c[(src[i] >> shift) & 0xFFFF]++;
                dst[p] = src[i]; vd[p] = vs[i];
            }
        }
        uint64_t* tk = src; src = dst; dst = tk;
        int32_t* tv = vs; vs = vd; vd = tv;
    }
    if (src != keys) { memcpy(keys, src, sizeof(uint64_t) * (size_t)n); memcpy(vals, vs, sizeof(int32_t) * (size_t)n); }
    free(k2); free(v2); free(cnt);
}

static inline uint64_t compose(int32_t contig, int32_t coord) {
    return ((uint64_t)(uint32_t)contig << 32) | (uint64_t)((uint32_t)coord ^ 0x80000000u);
}

/* ---- implicit augmented interval tree (the closest stand-in for the reference's index) -----------------
 * The reference joins through COITrees (coitrees 0.4.0 behind datafusion-bio-function-ranges,
 * /root/reference/Cargo.lock:1090-1094, docs/developers.md:629-639): the build intervals of a contig sorted by
 * start, laid out as a balanced binary tree, every node augmented with the maximum end of its subtree; a query
 * descends, pruning subtrees whose max end cannot reach the query and right subtrees whose starts lie past it.
 * Here the tree is implicit in the sorted array: the root of [lo, hi) is its midpoint (COITrees uses a van Emde
 * Boas layout of the same tree for cache behaviour; pruning rules and results are the same).  In-order traversal
 * emits the matches in (start, row) order, i.e. the order of orc_overlap_fast.                              */
static int32_t tree_build(orc_index* ix, int64_t lo, int64_t hi) {
    if (lo >= hi) return INT32_MIN;
    int64_t mid = lo + ((hi - lo) >> 1);
    int32_t m = ix->s_end[mid];
    int32_t l = tree_build(ix, lo, mid), r = tree_build(ix, mid + 1, hi);
    if (l > m) m = l;
    if (r > m) m = r;
    ix->tmax[mid] = m;
    return m;
}

/* visits the matches of one probe in order; emit == NULL only counts */
static int64_t tree_query(const orc_index* ix, int64_t a, int64_t b, int32_t qs, int32_t qe, int strict, int32_t pi,
                          int32_t* out_probe, int32_t* out_build, int64_t w, int64_t cap) {
    /* explicit stack of (lo, hi, state): state 0 = visit left subtree first, 1 = node itself then right subtree */
    int64_t st_lo[64], st_hi[64]; int st_s[64]; int sp = 0;
    int64_t found = 0;
    st_lo[0] = a; st_hi[0] = b; st_s[0] = 0; sp = 1;
    while (sp > 0) {
        int64_t lo = st_lo[sp - 1], hi = st_hi[sp - 1]; int state = st_s[sp - 1];
        --sp;
        if (lo >= hi) continue;
        int64_t mid = lo + ((hi - lo) >> 1);
        if (state == 0) {
            if (!cond_b(qs, ix->tmax[mid], strict)) continue;            /* nothing in this subtree ends late enough */
            st_lo[sp] = lo; st_hi[sp] = hi; st_s[sp] = 1; ++sp;          /* come back for the node and the right side */
            st_lo[sp] = lo; st_hi[sp] = mid; st_s[sp] = 0; ++sp;         /* left subtree first (in-order) */
        } else {
            if (!cond_a(ix->s_start[mid], qe, strict)) continue;         /* this start and all to the right are too late */
            if (cond_b(qs, ix->s_end[mid], strict)) {
                if (out_probe && w + found < cap) { out_probe[w + found] = pi; out_build[w + found] = ix->s_row[mid]; }
                ++found;
            }
            st_lo[sp] = mid + 1; st_hi[sp] = hi; st_s[sp] = 0; ++sp;
        }
    }
    return found;
}

orc_index* orc_index_build(const orc_side* build, int n_contigs) {
    orc_index* ix = (orc_index*)calloc(1, sizeof(orc_index));
    int64_t n = build->n;
    ix->n = n; ix->n_contigs = n_contigs;
    size_t nn = (size_t)(n > 0 ? n : 1);
    ix->seg = (int64_t*)calloc((size_t)n_contigs + 1, sizeof(int64_t));
    ix->s_start = (int32_t*)malloc(4 * nn); ix->s_end = (int32_t*)malloc(4 * nn);
    ix->s_row = (int32_t*)malloc(4 * nn);   ix->pmax = (int32_t*)malloc(4 * nn);
    ix->e_end = (int32_t*)malloc(4 * nn);   ix->e_pos = (int32_t*)malloc(4 * nn);
    uint64_t* keys = (uint64_t*)malloc(8 * nn);
    /* build rows whose contig id is outside [0, n_contigs) can never match:
     * they are parked in a trailing pseudo-segment by giving them the id n_contigs */
    int inverted = 0;
#pragma omp parallel for schedule(static) reduction(| : inverted) if (n >= (1 << 20))
    for (int64_t j = 0; j < n; ++j) {
        int32_t c = build->contig[j];
        if (c < 0 || c >= n_contigs) c = n_contigs;
        keys[j] = compose(c, build->start[j]);
        ix->s_row[j] = (int32_t)j;
        if (build->start[j] > build->end[j]) inverted |= 1;
    }
    ix->has_inverted = inverted;
    radix_sort_u64(keys, ix->s_row, n, 64);
    int64_t n_valid = n;
    for (int64_t p = 0; p < n; ++p) {
        int32_t c = (int32_t)(keys[p] >> 32);
        if (c >= n_contigs) { n_valid = p; break; }
        ix->seg[c + 1]++;
    }
    for (int c = 0; c < n_contigs; ++c) ix->seg[c + 1] += ix->seg[c];
#pragma omp parallel for schedule(static) if (n_valid >= (1 << 20))
    for (int64_t p = 0; p < n_valid; ++p) {
        int32_t r = ix->s_row[p];
        ix->s_start[p] = build->start[r];
        ix->s_end[p] = build->end[r];
    }
#pragma omp parallel for schedule(dynamic, 1) if (n_valid >= (1 << 20))
    for (int c = 0; c < n_contigs; ++c) {
        int32_t m = INT32_MIN;
        for (int64_t p = ix->seg[c]; p < ix->seg[c + 1]; ++p) {
            if (ix->s_end[p] > m) m = ix->s_end[p];
            ix->pmax[p] = m;
        }
    }
    /* ends sorted by (contig, end, position in start order) */
#pragma omp parallel for schedule(static) if (n_valid >= (1 << 20))
    for (int64_t p = 0; p < n_valid; ++p) {
        keys[p] = compose(build->contig[ix->s_row[p]], ix->s_end[p]);
        ix->e_pos[p] = (int32_t)p;
    }
    radix_sort_u64(keys, ix->e_pos, n_valid, 64);
#pragma omp parallel for schedule(static) if (n_valid >= (1 << 20))
    for (int64_t p = 0; p < n_valid; ++p) ix->e_end[p] = ix->s_end[ix->e_pos[p]];
    free(keys);
    ix->tmax = (int32_t*)malloc(4 * nn);
#pragma omp parallel for schedule(dynamic, 1) if (n_valid >= (1 << 20))
    for (int c = 0; c < n_contigs; ++c) tree_build(ix, ix->seg[c], ix->seg[c + 1]);
    return ix;
}

void orc_index_free(orc_index* ix) {
    if (!ix) return;
    free(ix->seg); free(ix->s_start); free(ix->s_end); free(ix->s_row);
    free(ix->pmax); free(ix->e_end); free(ix->e_pos); free(ix->tmax); free(ix);
}

/* first p in [lo,hi) with a[p] >= x */
static inline int64_t lower_bound32(const int32_t* a, int64_t lo, int64_t hi, int32_t x) {
    while (lo < hi) { int64_t m = lo + ((hi - lo) >> 1); if (a[m] < x) lo = m + 1; else hi = m; }
    return lo;
}
/* first p in [lo,hi) with a[p] > x */
static inline int64_t upper_bound32(const int32_t* a, int64_t lo, int64_t hi, int32_t x) {
    while (lo < hi) { int64_t m = lo + ((hi - lo) >> 1); if (a[m] <= x) lo = m + 1; else hi = m; }
    return lo;
}

/* hi = first position of the segment whose start fails "start (<) q.end" */
static inline int64_t bound_hi(const orc_index* ix, int64_t a, int64_t b, int32_t qe, int strict) {
    return strict ? lower_bound32(ix->s_start, a, b, qe) : upper_bound32(ix->s_start, a, b, qe);
}
/* lo = first position in [a,hi) whose prefix-max end satisfies "q.start (<) pmax" */
static inline int64_t bound_lo(const orc_index* ix, int64_t a, int64_t hi, int32_t qs, int strict) {
    return strict ? upper_bound32(ix->pmax, a, hi, qs) : lower_bound32(ix->pmax, a, hi, qs);
}
/* r = first position of the end-sorted segment whose end satisfies "q.start (<) end" */
static inline int64_t bound_r(const orc_index* ix, int64_t a, int64_t b, int32_t qs, int strict) {
    return strict ? upper_bound32(ix->e_end, a, b, qs) : lower_bound32(ix->e_end, a, b, qs);
}

static inline int seg_of(const orc_index* ix, int32_t c, int64_t* a, int64_t* b) {
    if (c < 0 || c >= ix->n_contigs) return 0;
    *a = ix->seg[c]; *b = ix->seg[c + 1];
    return *b > *a;
}

static inline int64_t count_one(const orc_index* ix, int32_t c, int32_t qs, int32_t qe, int strict) {
    int64_t a, b;
    if (!seg_of(ix, c, &a, &b)) return 0;
    int64_t hi = bound_hi(ix, a, b, qe, strict);
    int degenerate = ix->has_inverted || (strict ? qs >= qe : qs > qe);
    if (!degenerate) {
        /* count = #{b.start (<) q.end} - #{not q.start (<) b.end}: the formula
         * of the SQL sweep, polars_bio/range_op.py:548-595 */
        return (hi - a) - (bound_r(ix, a, b, qs, strict) - a);
    }
    int64_t lo = bound_lo(ix, a, hi, qs, strict), n = 0;
    for (int64_t p = lo; p < hi; ++p) n += cond_b(qs, ix->s_end[p], strict);
    return n;
}

void orc_count_overlaps_fast(const orc_index* ix, const orc_side* probe, int strict,
                             int64_t* counts, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    (void)threads;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < probe->n; ++i)
        counts[i] = count_one(ix, probe->contig[i], probe->start[i], probe->end[i], strict);
}

int64_t orc_overlap_fast(const orc_index* ix, const orc_side* probe, int strict,
                         int32_t* out_probe, int32_t* out_build, int64_t cap, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    int nt = omp_get_max_threads();
#else
    int nt = 1;
#endif
    (void)threads;
    int64_t np = probe->n;
    int64_t* part = (int64_t*)calloc((size_t)nt + 1, sizeof(int64_t));
    /* pass 1: per-thread totals over a static row partition */
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num();
#else
        int t = 0;
#endif
        int64_t lo_i = np * t / nt, hi_i = np * (t + 1) / nt, s = 0;
        for (int64_t i = lo_i; i < hi_i; ++i)
            s += count_one(ix, probe->contig[i], probe->start[i], probe->end[i], strict);
        part[t + 1] = s;
    }
    for (int t = 0; t < nt; ++t) part[t + 1] += part[t];
    int64_t total = part[nt];
    if (out_probe) {
        /* pass 2: fill at the scanned offsets */
#pragma omp parallel num_threads(nt)
        {
#ifdef _OPENMP
            int t = omp_get_thread_num();
#else
            int t = 0;
#endif
            int64_t lo_i = np * t / nt, hi_i = np * (t + 1) / nt, w = part[t];
            for (int64_t i = lo_i; i < hi_i; ++i) {
                int64_t a, b;
                if (!seg_of(ix, probe->contig[i], &a, &b)) continue;
                int32_t qs = probe->start[i], qe = probe->end[i];
                int64_t hi = bound_hi(ix, a, b, qe, strict);
                int64_t lo = bound_lo(ix, a, hi, qs, strict);
                for (int64_t p = lo; p < hi; ++p) {
                    if (cond_b(qs, ix->s_end[p], strict)) {
                        if (w < cap) { out_probe[w] = (int32_t)i; out_build[w] = ix->s_row[p]; }
                        ++w;
                    }
                }
            }
        }
    }
    free(part);
    return total;
}

static void nearest_one(const orc_index* ix, int32_t c, int32_t qs, int32_t qe, int strict,
                        int k, int include_overlaps,
                        int32_t* oi, int64_t* od, int32_t* on) {
    int32_t n = 0;
    int64_t a, b;
    for (int r = 0; r < k; ++r) { oi[r] = -1; od[r] = -1; }
    if (!seg_of(ix, c, &a, &b)) { *on = 0; return; }
    int64_t hi = bound_hi(ix, a, b, qe, strict);
    if (include_overlaps) {
        int64_t lo = bound_lo(ix, a, hi, qs, strict);
        for (int64_t p = lo; p < hi && n < k; ++p)
            if (cond_b(qs, ix->s_end[p], strict)) { oi[n] = ix->s_row[p]; od[n] = 0; ++n; }
    }
    /* left stream (class 1): rows with "start (<) q.end" that fail
     * "q.start (<) end", by end descending, runs of equal end in ascending
     * (start,row) order.  right stream (class 2): every row at or after hi
     * (fails "start (<) q.end"), by (start,row) ascending. */
    int64_t r_top = bound_r(ix, a, b, qs, strict);   /* rows failing B = e-order [a, r_top) */
    int64_t run_hi = r_top, run_lo = r_top, lp = r_top; /* current run [run_lo, run_hi), cursor lp */
    int64_t rp = hi;
    while (n < k) {
        for (;;) {
            /* skip rows of the run that are not class 1 (start fails A: they sit at or after hi) */
            while (lp < run_hi && ix->e_pos[lp] >= hi) ++lp;
            if (lp < run_hi || run_lo <= a) break;
            run_hi = run_lo;
            int32_t e = ix->e_end[run_hi - 1];
            run_lo = lower_bound32(ix->e_end, a, run_hi, e);
            lp = run_lo;
        }
        int have_l = lp < run_hi;
        int have_r = rp < b;
        if (!have_l && !have_r) break;
        int64_t dl = 0, dr = 0;
        if (have_l) { int64_t p = ix->e_pos[lp]; dl = gap_dist(qs, qe, ix->s_start[p], ix->s_end[p]); }
        if (have_r) dr = gap_dist(qs, qe, ix->s_start[rp], ix->s_end[rp]);
        if (have_l && (!have_r || dl <= dr)) {
            oi[n] = ix->s_row[ix->e_pos[lp]]; od[n] = dl; ++n; ++lp;
        } else {
            oi[n] = ix->s_row[rp]; od[n] = dr; ++n; ++rp;
        }
    }
    *on = n;
}

void orc_nearest_fast(const orc_index* ix, const orc_side* probe, int strict,
                      int k, int include_overlaps,
                      int32_t* out_idx, int64_t* out_dist, int32_t* out_n, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    (void)threads;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < probe->n; ++i)
        nearest_one(ix, probe->contig[i], probe->start[i], probe->end[i], strict, k,
                    include_overlaps, out_idx + i * k, out_dist + i * k, out_n + i);
}

int64_t orc_overlap_tree(const orc_index* ix, const orc_side* probe, int strict,
                         int32_t* out_probe, int32_t* out_build, int64_t cap, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    int nt = omp_get_max_threads();
#else
    int nt = 1;
#endif
    (void)threads;
    int64_t np = probe->n;
    int64_t* part = (int64_t*)calloc((size_t)nt + 1, sizeof(int64_t));
    for (int pass = 0; pass < (out_probe ? 2 : 1); ++pass) {
#pragma omp parallel num_threads(nt)
        {
#ifdef _OPENMP
            int t = omp_get_thread_num();
#else
            int t = 0;
#endif
            int64_t lo_i = np * t / nt, hi_i = np * (t + 1) / nt, w = pass ? part[t] : 0, s = 0;
            for (int64_t i = lo_i; i < hi_i; ++i) {
                int64_t a, b;
                if (!seg_of(ix, probe->contig[i], &a, &b)) continue;
                int64_t f = tree_query(ix, a, b, probe->start[i], probe->end[i], strict, (int32_t)i,
                                       pass ? out_probe : NULL, out_build, w, cap);
                s += f; w += f;
            }
            if (!pass) part[t + 1] = s;
        }
        if (!pass) for (int t = 0; t < nt; ++t) part[t + 1] += part[t];
    }
    int64_t total = part[nt];
    free(part);
    return total;
}

/* ---- timed CPU baseline: ONE call, ONE pass per probe row ------------------------------------------------
 * What a streaming CPU executor does (the reference's IntervalJoinExec queries its index once per probe row and
 * appends the matches to the current output batch, docs/developers.md:641-646): every thread walks its static
 * share of the probe rows once and appends (probe row, build row) pairs to its own output batch of BATCH pairs,
 * which is recycled when full (the consumer of a stream has taken it).  use_tree = 0: bound search + window scan
 * over the sorted arrays; 1: the implicit augmented interval tree.  sort_chunks != 0: every thread first sorts
 * its share by (contig, start) -- the sort is inside the timed call -- so that consecutive queries touch
 * neighbouring index rows.  Returns the number of pairs; *checksum = sum of the emitted build rows (compared with
 * the two-pass result by tests/test_oracle_golden.py). */
int64_t orc_overlap_baseline(const orc_index* ix, const orc_side* probe, int strict, int threads, int use_tree,
                             int sort_chunks, int64_t* checksum) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    int nt = omp_get_max_threads();
#else
    int nt = 1;
#endif
    (void)threads;
    enum { BATCH = 1 << 16 };
    const int64_t np = probe->n;
    int64_t total = 0, csum = 0;
#pragma omp parallel num_threads(nt) reduction(+ : total, csum)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num();
#else
        int t = 0;
#endif
        const int64_t lo_i = np * t / nt, hi_i = np * (t + 1) / nt, m = hi_i - lo_i;
        int32_t* op = (int32_t*)malloc(sizeof(int32_t) * BATCH);
        int32_t* ob = (int32_t*)malloc(sizeof(int32_t) * BATCH);
        int32_t* order = NULL;
        if (sort_chunks && m > 0) {
            uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)m);
            order = (int32_t*)malloc(sizeof(int32_t) * (size_t)m);
            for (int64_t k = 0; k < m; ++k) { keys[k] = compose(probe->contig[lo_i + k], probe->start[lo_i + k]); order[k] = (int32_t)(lo_i + k); }
            radix_sort_u64(keys, order, m, 48);
            free(keys);
        }
        int64_t w = 0;
        for (int64_t k = 0; k < m; ++k) {
            const int64_t i = order ? order[k] : lo_i + k;
            int64_t a, b;
            if (!seg_of(ix, probe->contig[i], &a, &b)) continue;
            const int32_t qs = probe->start[i], qe = probe->end[i];
            int64_t f = 0;
            if (use_tree) {
                f = tree_query(ix, a, b, qs, qe, strict, (int32_t)i, op, ob, w, BATCH);
                if (w + f > BATCH) {        /* batch full: hand it over, redo this row into the fresh one */
                    for (int64_t j = 0; j < w; ++j) csum += ob[j];
                    w = 0;
                    f = tree_query(ix, a, b, qs, qe, strict, (int32_t)i, op, ob, 0, BATCH);
                }
                w += f < BATCH ? f : BATCH;
            } else {
                const int64_t hi = bound_hi(ix, a, b, qe, strict);
                for (int64_t p = hi - 1; p >= a && cond_b(qs, ix->pmax[p], strict); --p) {
                    if (cond_b(qs, ix->s_end[p], strict)) {
                        if (w == BATCH) { for (int64_t j = 0; j < w; ++j) csum += ob[j]; w = 0; }
                        op[w] = (int32_t)i; ob[w] = ix->s_row[p]; ++w; ++f;
                    }
                }
            }
            total += f;
        }
        for (int64_t j = 0; j < w; ++j) csum += ob[j];
        free(op); free(ob); free(order);
    }
    if (checksum) *checksum = csum;
    return total;
}


/* Copy of one probe column whose pages are first touched by the thread that will read them in orc_overlap_baseline (the same
 * static shares np * t / nt): on a multi-socket host the all-core baseline then reads local memory instead of the one NUMA node
 * a single-threaded generator left the column on.  Not part of any timed region. */
void orc_place_i32(const int32_t* src, int32_t* dst, int64_t n, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    int nt = omp_get_max_threads();
#else
    int nt = 1;
#endif
    (void)threads;
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        if (hi > lo) memcpy(dst + lo, src + lo, sizeof(int32_t) * (size_t)(hi - lo));
    }
}


/* threads of the parallel regions that take their count from the runtime (the index build): a host that knows its CPU quota sets it */
void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
