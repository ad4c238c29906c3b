/*
 * ivj_oracle.h -- CPU restatement of polars-bio's interval-join semantics.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / the timed CPU baseline.
 *
 * The arithmetic of the reference lives in the un-vendored crate
 * datafusion-bio-function-ranges v0.11.0 (git 1ad88df6, Cargo.toml:65,
 * Cargo.lock:1829-1844; default index coitrees 0.4.0, Cargo.lock:1090-1094),
 * which is absent from /root/reference and cannot be built here (no
 * cargo/rustc).  The semantics restated here are therefore anchored on the
 * reference's own call sites, SQL restatement, tests and golden tables:
 *   - Strict/Weak mapping .......... polars_bio/range_op.py:56-84
 *   - overlap pair definition ...... tests/test_coordinate_system_metadata.py:738-819
 *   - count = two-rank formula ..... polars_bio/range_op.py:548-595 (SQL sweep)
 *   - count swap / df1 order ....... polars_bio/range_op.py:503-511, src/operation.rs:316-317
 *   - nearest side roles ........... src/operation.rs:143-158
 *   - nearest distance / tie-break . tests/_expected.py:130-172, docs/notebooks/tutorial.ipynb cell 13
 * Pinned by tests/test_oracle_golden.py against every golden table listed in
 * SURVEY.md section 8c.  What no reference test pins (nearest k>1, ties at
 * equal non-zero distance, overlap=False, absent contig) is marked
 * "parity unpinned" in DESIGN.md and follows the rule documented below.
 */
#ifndef IVJ_ORACLE_H
#define IVJ_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    const int32_t* contig; /* dictionary id of chrom, shared by both sides */
    const int32_t* start;
    const int32_t* end;
    int64_t n;
} orc_side;

/* filter_op: 0 = Weak (1-based closed, <=), 1 = Strict (0-based half-open, <)
 * -- src/option.rs:95-100. */

/* ---- brute force, O(Np*Nb): the definition itself --------------------- */

/* All pairs (i in probe, j in build) with equal contig and
 * probe.start[i] (<|<=) build.end[j] && build.start[j] (<|<=) probe.end[i].
 * Order: probe row ascending, then (build.start, build row) ascending.
 * Returns the number of pairs; writes at most cap of them. */
int64_t orc_overlap_brute(const orc_side* probe, const orc_side* build, int strict,
                          int32_t* out_probe, int32_t* out_build, int64_t cap);

/* counts[i] = |{j : overlap(probe[i], build[j])}| (Int64, range_op_helpers.py:316) */
void orc_count_overlaps_brute(const orc_side* probe, const orc_side* build, int strict,
                              int64_t* counts);

/* k nearest build rows for each probe row.
 * Candidate order (total): distance ascending; then class (0 overlapping;
 * 1 "left": build.start (<) probe.end holds but probe.start (<) build.end
 * fails; 2 "right": build.start (<) probe.end fails); then build.start, then
 * build row.  distance = 0 for overlapping pairs, else
 * max(build.start - probe.end, probe.start - build.end) (no +-1 correction:
 * tests/_expected.py:130-172 -> 34; tutorial cell 13 -> 1).
 * include_overlaps = 0 removes class 0.
 * out_idx / out_dist have Np*k slots; unused slots are -1.  A probe row whose
 * contig has no candidate gets zero filled slots (the Python layer turns that
 * into one null row).  out_n[i] = number of filled slots. */
void orc_nearest_brute(const orc_side* probe, const orc_side* build, int strict,
                       int k, int include_overlaps,
                       int32_t* out_idx, int64_t* out_dist, int32_t* out_n);

/* ---- sort + bound search: same answers, O((Np+Nb) log Nb + P) ---------
 * This is also the timed CPU baseline ("port" of the device algorithm).   */

typedef struct orc_index orc_index;
orc_index* orc_index_build(const orc_side* build, int n_contigs);
void orc_index_free(orc_index*);

/* counts via the two-rank formula with the exact-scan fallback for
 * degenerate rows; threads <= 0 -> omp default. */
void orc_count_overlaps_fast(const orc_index* ix, const orc_side* probe, int strict,
                             int64_t* counts, int threads);

/* Two-pass count -> exclusive scan -> fill, same output order as brute.
 * Pass out_probe = NULL to only count. Returns P. */
int64_t orc_overlap_fast(const orc_index* ix, const orc_side* probe, int strict,
                         int32_t* out_probe, int32_t* out_build, int64_t cap, int threads);

/* The same pairs in the same order through an implicit augmented interval tree over the sorted build side -- the
 * closest stand-in for the reference's COITrees index (pruning by subtree max end and by start); second timed CPU
 * baseline and a third independent implementation for the cross-checks. */
int64_t orc_overlap_tree(const orc_index* ix, const orc_side* probe, int strict,
                         int32_t* out_probe, int32_t* out_build, int64_t cap, int threads);

/* Timed CPU baseline: one call, one pass over the probe rows, matches appended to recycled per-thread batches
 * (see the .c file).  use_tree: 0 bound search + window scan, 1 interval tree; sort_chunks: every thread sorts
 * its share of the probe rows first (inside the timed call).  *checksum = sum of the emitted build rows. */
int64_t orc_overlap_baseline(const orc_index* ix, const orc_side* probe, int strict, int threads, int use_tree,
                             int sort_chunks, int64_t* checksum);

void orc_set_threads(int n);

/* copy of a probe column first-touched by the threads that read it in orc_overlap_baseline (NUMA placement; never timed) */
void orc_place_i32(const int32_t* src, int32_t* dst, int64_t n, int threads);

void orc_nearest_fast(const orc_index* ix, const orc_side* probe, int strict,
                      int k, int include_overlaps,
                      int32_t* out_idx, int64_t* out_dist, int32_t* out_n, int threads);

#ifdef __cplusplus
}
#endif
#endif
