/*
 * ivjoin.h -- C ABI of the MI355X-native interval-join engine (libivjoin_hip.so).
 *
 * This is the drop-in boundary for polars-bio's range-operation hot path.  It
 * replaces what the reference reaches through its PyO3 module
 * `polars_bio.polars_bio`:
 *
 *   range_operation_frame(py_ctx, df1, df2, range_options, limit)
 *       /root/reference/src/lib.rs:79-145   (Arrow streams in, joined frame out)
 *   do_range_operation -> do_overlap / do_nearest / do_count_overlaps_coverage_naive
 *       /root/reference/src/operation.rs:27-98, 202-304, 100-200, 306-350
 *   OverlapProvider / NearestProvider / CountOverlapsProvider (IntervalJoinExec + COITrees)
 *       third-party datafusion-bio-function-ranges v0.11.0, call sites
 *       /root/reference/src/operation.rs:146-158, 253-263, 331-340
 *   RangeOptions / FilterOp
 *       /root/reference/src/option.rs:6-41, 95-100
 * and, built on the same sorted index:
 *   the renaming SELECT over the joined batches = row materialisation   src/operation.rs:272-301
 *   do_merge / do_cluster / do_complement / do_subtract / coverage       src/operation.rs:352-510, 86-96
 *       (MergeProvider, ClusterProvider, ComplementProvider, SubtractProvider, CountOverlapsProvider(coverage))
 *   range_operation_lazy + the streaming scan (df1 as an Arrow C stream, lazy result batches, limit)
 *       src/lib.rs:154-214, src/scan.rs:294-357, polars_bio/range_op_io.py:31-174           -> ivj_stream_*
 *
 * Contract: plain pointers and sizes only.  The join keys cross the ABI as
 * three int32 columns per side: `contig` (dictionary id of the chrom string,
 * one dictionary shared by both sides, assigned by the caller), `start`, `end`.
 * Every other column of the user's frames is gathered by the returned row indices
 * (what the reference's SELECT in src/operation.rs:272-301 does with `left_*` /
 * `right_*`): fixed-width columns through HBM (ivj_take, ivj_take_dev), the rest on the host.
 *
 * Side roles (Appendix A of SURVEY.md): probe = df1 (streamed side of the
 * reference), build = df2 (indexed side) for all three operations, i.e. after
 * the swaps in polars_bio/range_op.py:511 and src/operation.rs:143-158.
 *
 * Limits: a side holds at most 0x7fff0000 rows (int32 row indices); the build side at most
 * 2^30 - n_contigs - 32 rows (int32 table offsets).  A context (ivj_ctx) is not thread-safe: use one
 * context per host thread / per device; different contexts may be used concurrently.
 *
 * All entry points return 0 on success, a negative IVJ_E* code otherwise;
 * ivj_last_error() returns the thread-local message of the last failure.
 * The library never retains a caller pointer after the call returns.
 */
#ifndef IVJOIN_H
#define IVJOIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IVJ_OK            0
#define IVJ_EINVAL       -1   /* bad argument */
#define IVJ_EHIP         -2   /* HIP runtime error (message has the hipError string) */
#define IVJ_ENOMEM       -3   /* device or host allocation failed */
#define IVJ_ECAPACITY    -4   /* caller-provided output capacity too small */
#define IVJ_ESTATE       -5   /* fill called without a matching count; index lacks what the call needs */
#define IVJ_EPEER        -6   /* multi-rank call: ANOTHER rank failed; this rank's part is complete, the result is not */

/* FilterOp of the reference (src/option.rs:95-100) */
#define IVJ_FILTER_WEAK   0   /* 1-based closed:    a.start <= b.end && b.start <= a.end */
#define IVJ_FILTER_STRICT 1   /* 0-based half-open: a.start <  b.end && b.start <  a.end */

typedef struct ivj_ctx ivj_ctx;      /* one device, one stream, scratch arena */
typedef struct ivj_index ivj_index;  /* sorted build side resident in HBM */

/* One side of the join: three int32 columns of length n.  Host pointers for
 * the host entry points, device pointers for the *_dev entry points.  Rows
 * whose contig id is outside [0, n_contigs) never match anything. */
typedef struct {
    const int32_t* contig;
    const int32_t* start;
    const int32_t* end;
    int64_t n;
    const int32_t* row_id;  /* optional: the row id to report for row i (global ids of a
                               contig shard); NULL = i.  Same memory space as the columns. */
} ivj_side;

/* The subset of RangeOptions (src/option.rs:6-41) that reaches the kernels. */
typedef struct {
    int32_t filter_op;         /* IVJ_FILTER_WEAK | IVJ_FILTER_STRICT */
    int32_t n_contigs;         /* size of the shared chrom dictionary */
    int32_t nearest_k;         /* RangeOptions.nearest_k, >= 1 (default 1) */
    int32_t include_overlaps;  /* RangeOptions.include_overlaps (default 1) */
    int32_t partition_mode;    /* how the probe side is ordered before the join kernels run.  0 auto: overlap (the fused pass and the
                                  count -> fill pair) takes the contig-aligned slice path (6) from 512 Ki probe rows against 64 Ki
                                  build rows on (round 6; rounds 4-5: 1.5 Mi x 256 Ki; dictionaries of <= 256 contigs; tools/policy_sweep.py), large inputs outside
                                  that and the per-probe kernels the 256-bucket path (1), small ones none (2); 1 256 genomic buckets + window-scan kernels (deterministic); 2 never (probe order);
                                  5 flat (256 buckets + load-balanced candidate test, ivj_overlap_fused_dev only; with 0 the
                                  fused entry point picks it by itself when capacity >= 16 pairs per probe row, i.e. for dense
                                  results); 6 LDS-resident index slices: one stable partition into <= 1536 slices of equal
                                  row count + join on the slice held in LDS (overlap count / fill / fused; falls back to 1
                                  when the build side exceeds 1536 x 5120 rows or for the other operations) */
    int32_t table_mode;        /* direct-address table form: 0 auto (16-byte records for build sides >= 2^20 rows; nearest k = 1 over
                                  64-byte LINES -- one fetch per probe, probes in input order -- for build sides >= 2^17 rows once
                                  the probe side is >= 8 x the build side), 1 records, 2 plain 4-byte bins, 3 records + the nearest
                                  lines whatever the sizes (128 bytes of index per build row; other operations: as 1) */
    int32_t slice_rows;        /* slice path: build rows per slice, 0 = auto (rows / 1024, rounded up to 64, <= 5120) */
    int32_t slice_chunk;       /* slice path: probes per work item of the join (round 6: the contig-aligned join's persistent workgroups draw groups of
                                  items and join a bucket's consecutive items as one run; the other slice joins: per workgroup), 0 = auto (multiple of 4096) */
    int32_t deterministic;     /* overlap count -> fill pair on the slice path: 1 = the output is identical from run to run (stable
                                  partition behind a histogram pass, +0.7 ms per 100 M probes); 0 = same pairs, the order of the probe rows inside a
                                  bucket tile may differ between runs (the reference leaves the row order unspecified) */
} ivj_opts;

/* Result of overlap on the host path: library-owned host buffers. */
typedef struct {
    int64_t n_pairs;
    int32_t* probe_idx;   /* row of df1 */
    int32_t* build_idx;   /* row of df2 */
} ivj_pairs;

/* Per-kernel timing of the last operation (HIP events on the ctx stream). */
typedef struct {
    char name[32];
    int32_t launches;
    float ms;             /* sum over launches */
} ivj_timing;

/* ---- library / context -------------------------------------------------- */
/* The structs of this header carry no size field: a host built against another revision of it would hand the library structs of
 * another length.  IVJ_ABI_VERSION is bumped whenever a struct or a signature changes (5: ivj_opts.deterministic, the lazy Arrow
 * entries, the per-probe all-gathers, ivj_host_scatter); a binding checks ivj_abi_version() == the IVJ_ABI_VERSION it was built
 * against right after loading the library (the ctypes binding does: polars_bio_amd/_engine.py::load_library; the Rust sketch in
 * INTEGRATION.md does) and refuses to run otherwise. */
#define IVJ_ABI_VERSION 5
int ivj_abi_version(void);
const char* ivj_last_error(void);
const char* ivj_version(void);
int ivj_device_count(int* n);
/* Host memory (bytes) a result allocation may use right now: MemAvailable of /proc/meminfo (page cache counts as
 * available), MemFree as fallback; the host entry points refuse a result larger than 7/8 of it with IVJ_ENOMEM. */
int64_t ivj_host_mem_available(void);
int ivj_ctx_create(int device, ivj_ctx** out);
void ivj_ctx_destroy(ivj_ctx* ctx);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream; NULL is the
 * HIP legacy default stream, which is what torch uses unless told otherwise).
 * (void*)-1 restores the context's own non-blocking stream. */
int ivj_ctx_set_stream(ivj_ctx* ctx, void* hip_stream);
int ivj_ctx_sync(ivj_ctx* ctx);
/* level 0: off; 1: HIP events around the probe kernels only; 2: around every kernel */
int ivj_ctx_enable_timing(ivj_ctx* ctx, int level);
/* Synchronises, then writes up to cap entries; *n = number of distinct kernels. */
int ivj_ctx_get_timings(ivj_ctx* ctx, ivj_timing* out, int cap, int* n);
/* Launches the empty kernel ivj::k_profile_mark on the context's stream: a step boundary a profiler's kernel trace /
 * counter collection can be cut at (bench.py attributes the PMC bytes of its own run to the timed steps with it). */
int ivj_ctx_profile_mark(ivj_ctx* ctx);

/* ---- host-buffer entry points (what the reference FFI would bind) ------- *
 * Inputs are borrowed host buffers; results come back in host memory.       */

/* pb.overlap: all (probe_row, build_row) pairs.  The pairs of one probe row are contiguous and
 * ordered by (build.start, build row).  Probe rows appear in input order (small inputs, partition_mode 2)
 * or bucket by bucket -- by genomic position of the probe end -- when the probe side was partitioned:
 * with partition_mode 1 (256 buckets) in the stable order of that partition; on the contig-aligned slice
 * path (the automatic choice from 512 Ki probe rows x 64 Ki build rows on) the default partition is the
 * UNORDERED sampled one, so the order of the probe rows inside a bucket (and with it the order of the
 * output) may differ from run to run while the pair SET is exact; opts->deterministic = 1 selects the
 * stable partition there and an output that is identical from run to run.  The reference leaves the row
 * order unspecified (every reference test sorts).  Free with ivj_pairs_free. */
int ivj_overlap(ivj_ctx* ctx, const ivj_side* probe, const ivj_side* build,
                const ivj_opts* opts, ivj_pairs* out);
void ivj_pairs_free(ivj_pairs* p);

/* pb.count_overlaps (naive_query=True): counts[i] for every probe row, probe
 * order kept (range_op_helpers.py:315-316: Int64). counts: caller buffer of probe->n. */
int ivj_count_overlaps(ivj_ctx* ctx, const ivj_side* probe, const ivj_side* build,
                       const ivj_opts* opts, int64_t* counts);

/* pb.nearest: for every probe row up to k build rows.  idx/dist: caller
 * buffers of probe->n * k (unused slots -1); n_found: probe->n. */
int ivj_nearest(ivj_ctx* ctx, const ivj_side* probe, const ivj_side* build,
                const ivj_opts* opts, int32_t* idx, int64_t* dist, int32_t* n_found);

/* ---- device-resident entry points (inputs and outputs stay in HBM) ------ *
 * Used by bench.py, the multi-GPU driver and any caller that already holds  *
 * the columns on the GPU.  All work is enqueued on the context stream.      */

/* Sort the build side by (contig, start), derive segment offsets and the
 * prefix-max-of-end array.  with_end_order != 0 additionally sorts the ends
 * (needed by count_overlaps and by nearest with k > 1 or include_overlaps = 0;
 * built on demand otherwise); bit 1 (value 2): sweep-only index for ivj_merge_dev / ivj_cluster_dev -- the
 * lookup tables of the join kernels are not built (every other *_dev call then fails with IVJ_ESTATE).
 * The index copies what it needs: the caller's build columns may be released after the call returns. */
int ivj_index_build_dev(ivj_ctx* ctx, const ivj_side* build_dev, const ivj_opts* opts,
                        int with_end_order, ivj_index** out);
void ivj_index_free(ivj_index* ix);

/* overlap, two calls: count (+ scan, returns the number of pairs after one
 * 8-byte D2H) then fill into caller-allocated device buffers of that size.
 * The per-probe state between the two calls lives in ctx. */
int ivj_overlap_count_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev,
                          const ivj_opts* opts, int64_t* n_pairs);
int ivj_overlap_fill_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev,
                         const ivj_opts* opts, int32_t* probe_idx_dev, int32_t* build_idx_dev,
                         int64_t capacity);

/* overlap in ONE pass for callers that bring an output buffer of known capacity (steady-state /
 * streaming batches: the previous batch sized it): counting and emitting are fused, every tile
 * reserves its output range with one atomic, nothing is written to or re-read from HBM in between.
 * *n_pairs receives the total.  Returns IVJ_ECAPACITY (and writes nothing past the capacity) when
 * the buffer is too small: grow it to *n_pairs and call again, or use the count/fill pair.
 * The pairs of one probe row are contiguous and ordered; the order of the tiles is NOT
 * reproducible from run to run (the count/fill pair is the deterministic path). */
int ivj_overlap_fused_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts,
                          int32_t* probe_idx_dev, int32_t* build_idx_dev, int64_t capacity, int64_t* n_pairs);

int ivj_count_overlaps_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev,
                           const ivj_opts* opts, int64_t* counts_dev);

int ivj_nearest_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev,
                    const ivj_opts* opts, int32_t* idx_dev, int64_t* dist_dev, int32_t* n_found_dev);

/* ---- row materialisation -------------------- *
 * The step right after the join: gather the columns of both sides for every emitted pair     *
 * (reference: the renaming SELECT of src/operation.rs:272-301 over the joined batches).      *
 * Output columns are Arrow-layout value buffers (fixed-width little-endian values, optional   *
 * validity bitmap, LSB first).                                                               */

/* Key columns of an overlap result, n_pairs values each.  Device pointers for
 * ivj_materialize_dev (caller-allocated; a NULL column is skipped), library-owned host
 * buffers for ivj_overlap_rows (free with ivj_rows_free). */
typedef struct {
    int64_t n_pairs;
    int32_t* probe_idx;   /* row of df1 */
    int32_t* build_idx;   /* row of df2 */
    int32_t* contig;      /* contig id of the pair (equal on both sides) */
    int32_t* start_1;     /* df1.start[probe_idx] */
    int32_t* end_1;
    int32_t* start_2;     /* df2.start[build_idx] */
    int32_t* end_2;
} ivj_rows;

/* Gathers contig / start_1 / end_1 / start_2 / end_2 for the pairs (rows->probe_idx,
 * rows->build_idx) = positions into probe_dev / build_dev, in one pass over the pair list. */
int ivj_materialize_dev(ivj_ctx* ctx, const ivj_side* probe_dev, const ivj_side* build_dev, const ivj_rows* rows);

/* Join AND materialisation in one pass (the fast path: the probe values never leave the workgroup,
 * the build values come with one 16-byte read per row; a separate ivj_materialize_dev over the
 * finished pair list has to fetch the probe columns at random).  rows_dev: device column pointers
 * (NULL = column not wanted), rows_dev->n_pairs = capacity of every column on entry.  Otherwise the
 * contract of ivj_overlap_fused_dev: *n_pairs = total, IVJ_ECAPACITY when it does not fit, rows of one
 * probe contiguous and ordered, tile order not reproducible.  Works with row_id (global ids). */
int ivj_overlap_fused_rows_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts,
                               const ivj_rows* rows_dev, int64_t* n_pairs);

/* Arrow `take` of one fixed-width device column: dst[i] = src[idx[i]]; elem_bytes is 4 or 8.
 * A negative index (the "no candidate" slot of nearest) writes 0 and, when validity_dev is given
 * (ceil(n / 64) 64-bit words), clears that row's validity bit. */
int ivj_take_dev(ivj_ctx* ctx, const void* src_dev, int32_t elem_bytes, const int32_t* idx_dev, int64_t n,
                 void* dst_dev, uint64_t* validity_dev);

/* Host-buffer form of the same `take` for several columns that share ONE index column (the non-key columns of a joined
 * frame: src/operation.rs:272-301 gathers every column of both sides for every pair).  idx (n int32, host) is uploaded
 * once; every column c -- src[c]: src_rows[c] values of elem_bytes[c] = 4 or 8 bytes, host -- is uploaded, gathered in
 * HBM and downloaded into the caller's dst[c] (n values); validity[c] (may be NULL; ceil(n / 64) words) receives the
 * Arrow validity bitmap of the negative-index slots.  The copies go through the context's pinned staging slots (never through
 * hipHostRegister or a pageable hipMemcpy of the caller's pointers); dst is first-touched by a few threads.
 * idx values must be < src_rows[c]. */
int ivj_take(ivj_ctx* ctx, const int32_t* idx, int64_t n, int32_t n_cols, const void* const* src, const int64_t* src_rows,
             const int32_t* elem_bytes, void* const* dst, uint64_t* const* validity);

/* Host-buffer form of overlap + materialisation: index pairs AND the five key columns come back
 * (one H2D of the inputs, join and gathers in HBM, one D2H per column). */
int ivj_overlap_rows(ivj_ctx* ctx, const ivj_side* probe, const ivj_side* build, const ivj_opts* opts, ivj_rows* out);
void ivj_rows_free(ivj_rows* rows);

/* Arrow C Data Interface export of an ivj_overlap_rows result as ONE struct array
 * {probe_idx, build_idx, contig, start_1, end_1, start_2, end_2 : int32 not null} without copying:
 * ownership of the host buffers moves to the ArrowArray (its release callback frees them; *rows is
 * cleared).  out_array / out_schema point to caller-allocated `struct ArrowArray` / `struct
 * ArrowSchema` (https://arrow.apache.org/docs/format/CDataInterface.html); a consumer such as
 * pyarrow.RecordBatch._import_from_c takes it from there. */
int ivj_rows_export_arrow(ivj_rows* rows, void* out_array, void* out_schema);

/* ---- sort-scan family: merge / cluster / coverage ---------------- *
 * One sweep over the (contig, start)-sorted index with its prefix max of the ends: a row joins the  *
 * running cluster iff start (<) max end so far + min_dist, (<) = "<" Strict / "<=" Weak.             *
 * Replaces MergeProvider / ClusterProvider / CountOverlapsProvider(coverage = true)               *
 * (src/operation.rs:352-418, 306-350).  min_dist = 0 does not merge bookended half-open intervals   *
 * (tests/_expected.py:174-181).                                                                   */

/* pb.merge result on the host path: library-owned buffers in (contig id, start) order. */
typedef struct {
    int64_t n;
    int32_t* contig;        /* contig id; -1 for the pseudo-contig of rows outside the dictionary */
    int32_t* start;
    int32_t* end;
    int64_t* n_intervals;   /* rows merged into the interval */
} ivj_merged;

int ivj_merge(ivj_ctx* ctx, const ivj_side* frame, const ivj_opts* opts, int64_t min_dist, ivj_merged* out);
void ivj_merged_free(ivj_merged* m);
/* pb.cluster: per input row the cluster id (clusters numbered in (contig id, start) order) and the
 * cluster's bounds; caller buffers of frame->n. */
int ivj_cluster(ivj_ctx* ctx, const ivj_side* frame, const ivj_opts* opts, int64_t min_dist, int64_t* cluster,
                int32_t* cluster_start, int32_t* cluster_end, int64_t* n_clusters);
/* pb.coverage: for every probe row the number of its positions covered by the union of the build
 * intervals of the same contig (Int64, probe order kept; [s, e) Strict, [s, e] Weak). */
int ivj_coverage(ivj_ctx* ctx, const ivj_side* probe, const ivj_side* build, const ivj_opts* opts, int64_t* coverage);

/* pb.subtract / pb.complement (SubtractProvider / ComplementProvider, src/operation.rs:420-510): every left
 * interval minus the union of the right intervals of its contig.  Result = the remaining pieces as
 * (left row, start, end), left-row order, ascending inside a row; a fully covered row yields nothing, a row
 * whose contig is absent on the right comes back whole.  Library-owned host buffers. */
typedef struct {
    int64_t n;
    int32_t* row;     /* row of the left side (left->row_id when given) */
    int32_t* start;
    int32_t* end;
} ivj_pieces;

int ivj_subtract(ivj_ctx* ctx, const ivj_side* left, const ivj_side* right, const ivj_opts* opts, ivj_pieces* out);
/* complement = the gaps of `frame` inside every interval of `view` (e.g. one row per chromosome):
 * ivj_subtract(view, frame); row = view row. */
int ivj_complement(ivj_ctx* ctx, const ivj_side* frame, const ivj_side* view, const ivj_opts* opts, ivj_pieces* out);
void ivj_pieces_free(ivj_pieces* p);

/* device-resident forms: the index of the frame / build side is built with ivj_index_build_dev */
/* subtract with the RIGHT side indexed; *n_pieces = total, IVJ_ECAPACITY when it exceeds the capacity of the
 * caller's columns (nothing is written then: grow and call again). */
int ivj_subtract_dev(ivj_ctx* ctx, ivj_index* right_ix, const ivj_side* left_dev, const ivj_opts* opts, int64_t capacity,
                     int32_t* row_dev, int32_t* start_dev, int32_t* end_dev, int64_t* n_pieces);
int ivj_cluster_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_opts* opts, int64_t min_dist, int64_t* cluster_dev,
                    int32_t* cluster_start_dev, int32_t* cluster_end_dev, int64_t* n_clusters);
int ivj_merge_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_opts* opts, int64_t min_dist, int64_t capacity,
                  int32_t* contig_dev, int32_t* start_dev, int32_t* end_dev, int64_t* n_intervals_dev, int64_t* n_merged);
int ivj_coverage_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, int64_t* coverage_dev);

/* Arrow C Data Interface IMPORT of one side (zero copy): `array` / `schema` describe a struct array (or record
 * batch) whose children named contig / start / end (any order, other children ignored) are int32 without nulls --
 * what the reference hands its executor as an ArrowArrayStream batch after the chrom column has been dictionary
 * encoded (src/lib.rs:89-100 collects the stream; range_op_io.py:398-418 builds it).  Fills *out with pointers INTO
 * the Arrow buffers (offset applied); nothing is copied and nothing is released: the caller keeps the ArrowArray
 * alive for as long as *out is used.  Needs no device. */
int ivj_side_from_arrow(const void* array, const void* schema, ivj_side* out);

/* ---- streaming probe side ----------------------------------------------------- *
 * The reference streams df1 through its executor as an Arrow C stream and yields result batches lazily            *
 * (polars_bio/range_op_io.py:100-174, src/lib.rs:154-214, fan-out with back-pressure src/scan.rs:294-357).         *
 * Here the build side (df2) is indexed once and stays in HBM; the probe side is SUBMITTED batch by batch (e.g. the *
 * record batches of an ArrowArrayStream, columns viewed with ivj_side_from_arrow).  Every submit overlaps the H2D   *
 * copy of the batch it was given (pinned staging slot, copy stream), the join of the batch before it (compute       *
 * stream) and the D2H copy of the batch before that (pinned result slot, copy-back stream), and hands out the        *
 * results of the batch submitted two calls earlier.  ivj_stream_flush drains what is left.                            */
#define IVJ_STREAM_OVERLAP  0   /* pairs (probe row of the batch, build row)                  */
#define IVJ_STREAM_COUNT    1   /* count_overlaps: int64 count per probe row of the batch      */
#define IVJ_STREAM_NEAREST  2   /* nearest: opts->nearest_k build rows, distances, n_found     */

typedef struct ivj_stream ivj_stream;

/* Results of ONE earlier batch; the buffers are pinned host memory owned by the stream, valid until the next call
 * on the stream.  batch = -1: nothing was ready. */
typedef struct {
    int64_t batch;          /* 0-based index of the submitted batch these results belong to, -1 = none      */
    int64_t n_probe;        /* probe rows of that batch                                                     */
    int64_t n;              /* overlap: pairs; count / nearest: n_probe                                     */
    int32_t* probe_idx;     /* overlap: probe row INSIDE the batch                                          */
    int32_t* build_idx;     /* overlap: build row; nearest: n_probe * k build rows (-1 = no candidate)      */
    int64_t* counts;        /* count_overlaps                                                               */
    int64_t* dist;          /* nearest: n_probe * k distances                                               */
    int32_t* n_found;       /* nearest: filled slots per probe row                                          */
} ivj_stream_result;

/* build: host columns (borrowed for the call).  max_batch_rows bounds the probe batches (pinned staging is sized by it). */
int ivj_stream_open(ivj_ctx* ctx, const ivj_side* build, const ivj_opts* opts, int op, int64_t max_batch_rows, ivj_stream** out);
/* batch: host columns of the next probe batch (borrowed for the call; copied into the pinned staging slot). */
int ivj_stream_submit(ivj_stream* st, const ivj_side* batch, ivj_stream_result* done);
/* no more input: call until done->batch == -1 */
int ivj_stream_flush(ivj_stream* st, ivj_stream_result* done);
void ivj_stream_close(ivj_stream* st);

/* ---- multi-GPU: one rank per GPU, contig sharding, all-gatherv of the result batches over RCCL (xGMI) ---------------- *
 * Replaces the reference's only parallelism -- DataFusion target_partitions over probe rows (src/scan.rs:233-277,
 * polars_bio/context.py:36) -- for a host that drives several GPUs: intervals on different contigs never interact
 * (the reference builds one tree per contig, range_op.py:550), so every rank joins the rows of ITS contigs (global row
 * ids in ivj_side.row_id) with no collective on the data path, and the variable-length results are exchanged with one
 * ncclAllGather of the counts + ONE grouped batch of ncclSend / ncclRecv (every GPU pair on its own xGMI link).
 * RCCL is loaded on first use (librccl.so.1); nothing here needs PyTorch.  A communicator of world 1 never touches RCCL --
 * unless IVJ_COMM_NO_SHORTCUT=1 is set when it is created (round 6; tests): it is then a REAL communicator (ncclGetUniqueId +
 * ncclCommInitRank), its count all-gather an ncclAllGather and its own slice travels through a grouped ncclSend + ncclRecv to
 * itself, so that the RCCL branch of every call below runs on a 1-GPU box (tests/test_comm.py::test_real_rccl_on_one_rank_*). */
typedef struct ivj_comm ivj_comm;
#define IVJ_UNIQUE_ID_BYTES 128
/* rank 0 makes the id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by any channel the host has */
int ivj_comm_unique_id(void* id_out);
/* one process per GPU: ncclCommInitRank on the context's device (collective: every rank calls it) */
int ivj_comm_create(ivj_ctx* ctx, const void* unique_id, int rank, int world, ivj_comm** out);
/* one process, one context per device: ncclCommInitAll; out[i] = communicator of ctxs[i] (rank i).  Calls on the n
 * communicators that belong together (ivj_allgather_counts, ivj_allgatherv_dev, ivj_overlap_allgather_dev) must then
 * come from n host threads, one per rank.  Contexts that SHARE a device (RCCL wants one device per rank), or
 * IVJ_COMM_LOOPBACK=1, get the library's in-process transport instead of RCCL: a host-memory rendezvous + device copies
 * issued by the receiving rank, same calls, same order, same blocking behaviour (a rank that never arrives makes its
 * peers fail with IVJ_EHIP after IVJ_COMM_LOOPBACK_TIMEOUT seconds, default 120, instead of waiting forever) -- a 1-GPU
 * box runs the multi-rank protocol with it. */
int ivj_comm_create_local(ivj_ctx* const* ctxs, int n, ivj_comm** out);
void ivj_comm_destroy(ivj_comm* comm);
int ivj_comm_info(const ivj_comm* comm, int* rank, int* world);
/* counts[r] = n_local of rank r, on every rank (host array of `world` entries) */
int ivj_allgather_counts(ivj_comm* comm, int64_t n_local, int64_t* counts);
/* all-gatherv of n_cols device columns of elem_bytes-wide elements: recv_cols[k] (capacity = sum of counts) receives
 * the concatenation over ranks, in rank order, of every rank's send_cols[k][0 .. counts[rank]).  Ordered after the work
 * queued on the context's stream; returns when the exchange is complete. */
int ivj_allgatherv_dev(ivj_comm* comm, const void* const* send_cols, void* const* recv_cols, int n_cols, int elem_bytes,
                       const int64_t* counts);
/* pb.overlap of this rank's shard + the all-gatherv of the pairs, with the exchange OVERLAPPING the join: the rank's
 * probe rows are cut into n_chunks contiguous chunks (the same number on every rank, 1 .. 64); while chunk i is joined
 * (fused single pass) a helper thread exchanges chunk i - 1 straight into the caller's columns.  Every rank ends up with
 * all *n_total pairs (layout: chunk after chunk, inside a chunk rank after rank; the pairs of one probe row contiguous);
 * *n_local = this rank's share.
 * Failure contract (no rank is left waiting in a collective, whatever happens on another one):
 *   - every rank takes part in the count all-gather of EVERY chunk.  A rank whose own work fails (join, allocation) marks
 *     the failed chunk and all later ones as failed instead of joining them and returns its own error code after the last
 *     chunk; every other rank completes its chunks and returns IVJ_EPEER (the failed rank's pairs from that chunk on are
 *     missing from the columns, *n_total counts what was exchanged);
 *   - IVJ_ECAPACITY on EVERY rank when the pairs outgrow the capacity of ANY rank: the decision is taken from the gathered
 *     counts, so all ranks stop moving pairs at the same chunk (nothing is written past a capacity), the remaining chunks
 *     are still joined and counted, and *n_total = the capacity the call needs -- one regrow step is enough;
 *   - an error of the collective itself (RCCL, the exchange stream) is returned as IVJ_EHIP; nothing can be promised to
 *     the peers then.
 * IVJ_FAULT_ALLGATHER="<rank>:<chunk>" in the environment makes that chunk's join on that rank fail (test knob). */
int ivj_overlap_allgather_dev(ivj_comm* comm, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, int n_chunks,
                              int32_t* probe_idx_dev, int32_t* build_idx_dev, int64_t capacity, int64_t* n_total, int64_t* n_local);

/* pb.count_overlaps / pb.nearest of this rank's shard + the exchange of the PER-PROBE results ("the gather is
 * of fixed-width per-probe results scattered back to original probe order (carry probe row id)"; replaces the single-process
 * providers behind src/operation.rs:331-340 and :146-158 for a multi-GPU host).  probe_dev->row_id = the GLOBAL probe rows of the
 * shard (required when world > 1; NULL at world 1 means 0 .. n-1), n_total = probe rows of the whole job (the same on every rank).
 * Every rank ends up with the full-length columns in global probe order: counts_dev[n_total] (int64), or idx_dev[n_total * k] /
 * dist_dev[n_total * k] / n_found_dev[n_total] with k = opts->nearest_k.  Rows no rank reports keep count 0 / build row -1, distance -1,
 * n_found 0.  On the wire: {row int32, count int32} (8 bytes per probe; counts are bounded by the build rows) and
 * {row int32, k x int32, k x int64, int32}: ONE count all-gather + ONE grouped send / receive batch.  The row column on the wire IS
 * probe_dev->row_id and the count column comes straight out of the shard's kernel (no pack pass); this rank's own slice is read where
 * it lies.  Receiver (round 6): a shard whose global rows ASCEND -- what ivj_host_shard and every host that keeps df1's order produce --
 * is MERGED: the rows of an output tile are one contiguous segment of every sender's columns (bound search per (sender, tile)), read
 * coalesced, placed by row in LDS, written coalesced with the defaults filled in (config 5 through this call on one rank: 10.6 -> 2.8 ms).
 * Senders in any other order are detected by the merge's own checks (every placed row belongs to its tile, the placed rows are counted)
 * and take the round-5 form -- defaults, then one store per reported row.  A row id reported twice is IVJ_EINVAL (merge form; the
 * scatter form keeps the later store), a row id outside [0, n_total) IVJ_EINVAL in both.
 * Failure contract: every rank reaches the count all-gather whatever happened to its own shard; a rank whose work failed returns its
 * own error, every other rank IVJ_EPEER, and nothing is sent or received (decided from the gathered values, so nobody waits in a
 * collective); ranks that disagree on n_total, or report more rows than n_total, get IVJ_EINVAL on every rank.
 * IVJ_FAULT_ALLGATHER="<rank>:0" makes that rank's shard fail (test knob). */
int ivj_count_overlaps_allgather_dev(ivj_comm* comm, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, int64_t n_total,
                                     int64_t* counts_dev);
int ivj_nearest_allgather_dev(ivj_comm* comm, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, int64_t n_total,
                              int32_t* idx_dev, int64_t* dist_dev, int32_t* n_found_dev);

/* ---- the one-call Arrow entry: two ArrowArrayStreams in, a stream of joined record batches out -------------------------- *
 * This is the shape of the reference's own FFI: range_operation_frame / range_operation_lazy take df1 and df2 as Arrow C
 * streams with a STRING chrom and start / end of any integer width and answer with a lazy frame of joined rows
 * (/root/reference/src/lib.rs:79-145, 154-214; the renaming SELECT over the joined batches: src/operation.rs:272-301,
 * 170-197).  A host binds ONE call per operation and re-implements nothing: both streams are drained (and released), chrom
 * (utf8 / large_utf8 / a dictionary of those, per batch) is encoded with one dictionary over both sides, start / end
 * (int8 .. int64, signed or unsigned) are narrowed to int32 with the reference's range check
 * (docs/features/operations.md:36-37; nulls in a coordinate column and values beyond int32 are IVJ_EINVAL with the column's
 * name), the join runs on the device through the host entry points above, and `out_stream` -- a caller-allocated
 * `struct ArrowArrayStream` (https://arrow.apache.org/docs/format/CStreamInterface.html) -- yields the result in batches of
 * batch_rows rows (<= 0: 1 Mi) that are assembled WHEN PULLED: every column of df1 named <name><suffix1>, then every
 * column of df2 named <name><suffix2>, gathered by the pair indices (fixed-width primitives of 1 .. 32 bytes, bool,
 * utf8 / large_utf8 / binary / large_binary; dictionary-encoded columns are delivered decoded to their value type; any
 * other column type is refused by name).  cols1 / cols2: {chrom, start, end} column names, NULL = those defaults.
 * opts->n_contigs is ignored (the dictionary is made inside).  limit >= 0 bounds the result rows (src/lib.rs:80-88, 125-130).
 * The stream pointers are `struct ArrowArrayStream*` (void* here so that this header needs no Arrow declarations). */
int ivj_overlap_arrow_stream(ivj_ctx* ctx, void* df1_stream, void* df2_stream, const char* const* cols1, const char* const* cols2,
                             const ivj_opts* opts, const char* suffix1 /* NULL: "_1" */, const char* suffix2 /* NULL: "_2" */,
                             int64_t batch_rows, int64_t limit, void* out_stream);
/* df1 columns (<name><suffix1>, NULL: no suffix) + count: int64, df1 row order (range_op.py:418-511, src/operation.rs:306-350) */
int ivj_count_overlaps_arrow_stream(ivj_ctx* ctx, void* df1_stream, void* df2_stream, const char* const* cols1, const char* const* cols2,
                                    const ivj_opts* opts, const char* suffix1, int64_t batch_rows, int64_t limit, void* out_stream);
/* per df1 row its opts->nearest_k nearest df2 rows (one result row each; a df1 row without any keeps one row with null df2
 * columns), + distance: int64 when with_distance (src/operation.rs:100-200) */
int ivj_nearest_arrow_stream(ivj_ctx* ctx, void* df1_stream, void* df2_stream, const char* const* cols1, const char* const* cols2,
                             const ivj_opts* opts, const char* suffix1, const char* suffix2, int32_t with_distance, int64_t batch_rows,
                             int64_t limit, void* out_stream);

/* The LAZY forms (round 5) -- /root/reference/src/lib.rs:154-214 range_operation_lazy, src/scan.rs:294-357, polars_bio/range_op_io.py:100-174:
 * df2 (the build side) is drained, encoded and indexed once; df1 stays a STREAM and is pulled batch by batch from inside the result
 * stream's get_next: a batch is encoded with the session's chrom dictionary, narrowed to int32, submitted to a streaming probe session
 * (ivj_stream_*: its H2D copy overlaps the join of the batch before it and the D2H copy of the batch before that), and the results that
 * come back -- those of the batch submitted two turns earlier -- are assembled into record batches of batch_rows rows.  df1 batches
 * above max_batch_rows (<= 0: 4 Mi; it sizes the session's pinned staging) are submitted in slices; batches BELOW min(max_batch_rows,
 * 2 Mi) rows are coalesced: the library pulls until it holds that many rows (or df1 ends) and treats the group as one batch (a group
 * never exceeds twice that size unless a single batch does).  Host memory: df2 + three such groups + one group's result, whatever the
 * length of df1 (which may pass 2^31 rows).  Result rows come in df1 order.  limit >= 0: df1 is pulled ONE batch at a time (no
 * coalescing: the call may need very little of it) and not at all after the limit is reached.  Errors of a later batch (a coordinate beyond int32, a malformed batch) surface from
 * get_next (errno, text from get_last_error) -- as the reference's surface at collect time.  OWNERSHIP (both forms, eager and lazy): the
 * two input streams are consumed by the call whatever its outcome -- on every error path their release callbacks have run (the lazy
 * form keeps df1 until the result stream is exhausted or released).  The result stream works on `ctx` whenever it is pulled: pull it
 * from one thread at a time and not concurrently with other calls on the same context; release it before the context. */
int ivj_overlap_arrow_stream_lazy(ivj_ctx* ctx, void* df1_stream, void* df2_stream, const char* const* cols1, const char* const* cols2,
                                  const ivj_opts* opts, const char* suffix1, const char* suffix2, int64_t batch_rows, int64_t max_batch_rows,
                                  int64_t limit, void* out_stream);
int ivj_count_overlaps_arrow_stream_lazy(ivj_ctx* ctx, void* df1_stream, void* df2_stream, const char* const* cols1, const char* const* cols2,
                                         const ivj_opts* opts, const char* suffix1, int64_t batch_rows, int64_t max_batch_rows, int64_t limit,
                                         void* out_stream);
int ivj_nearest_arrow_stream_lazy(ivj_ctx* ctx, void* df1_stream, void* df2_stream, const char* const* cols1, const char* const* cols2,
                                  const ivj_opts* opts, const char* suffix1, const char* suffix2, int32_t with_distance, int64_t batch_rows,
                                  int64_t max_batch_rows, int64_t limit, void* out_stream);

/* The two host halves of that call on their own (no device, no context): */
/* ... the key columns the join sees: both streams drained (and released), chrom encoded, coordinates narrowed */
typedef struct {
    int64_t n1, n2;
    int32_t *contig1, *start1, *end1;      /* n1 values each */
    int32_t *contig2, *start2, *end2;      /* n2 values each */
    int32_t n_contigs;
    int64_t* name_offsets;                 /* n_contigs + 1: name v of the shared dictionary = name_bytes[name_offsets[v] .. name_offsets[v + 1]) */
    char* name_bytes;
} ivj_arrow_keys;
int ivj_arrow_encode_keys(void* df1_stream, void* df2_stream, const char* const* cols1, const char* const* cols2, ivj_arrow_keys* out);
void ivj_arrow_keys_free(ivj_arrow_keys* keys);
/* ... the row assembly: rows idx[0 .. n) (negative or out of range: a null row) of the drained stream as a new stream */
int ivj_arrow_take_stream(void* in_stream, const int64_t* idx, int64_t n, int64_t batch_rows, void* out_stream);


/* ---- device memory helpers for callers without a HIP binding ------------ */
int ivj_dev_alloc(ivj_ctx* ctx, int64_t bytes, void** out);
int ivj_dev_free(ivj_ctx* ctx, void* p);
int ivj_memcpy_h2d(ivj_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes);
int ivj_memcpy_d2h(ivj_ctx* ctx, void* dst_host, const void* src_dev, int64_t bytes);

/* ---- host-side helpers of the front door (no device work, no context; plain std::thread workers; threads <= 0: 32) ----------
 * What the reference's executor does around the join in Rust -- the dictionary handling of the chrom key and the column gathers
 * of the renaming SELECT (src/operation.rs:272-301), the int32 coordinate limit (docs/features/operations.md:36-37) -- as ONE
 * pass over the rows each; the Python front door (polars_bio_amd/_arrow.py) calls them on the buffers of its Arrow columns. */

/* Coordinate column of src_bytes-wide (1, 2, 4, 8) signed / unsigned integers -> int32, with the column's minimum and maximum
 * (the caller refuses a column that leaves the int32 range; unsigned values beyond INT64_MAX are reported as INT64_MAX). */
int ivj_host_narrow_i32(const void* src, int32_t src_bytes, int32_t is_unsigned, int64_t n, int32_t* dst, int64_t* out_min, int64_t* out_max,
                        int32_t threads);

/* Arrow string / large_string column (n + 1 offsets of offset_bytes = 4 / 8, the value bytes, optional validity bitmap read from
 * bit validity_bit0) -> ids[n] (dictionary ids in first-occurrence order, -1 for a null) and dict_rows[*n_values] (one row that
 * holds each value).  IVJ_ECAPACITY: more than dict_cap (or 4096) distinct values -- the caller falls back to its own encoder. */
int ivj_host_encode_utf8(const void* offsets, int32_t offset_bytes, const uint8_t* data, const uint8_t* validity, int64_t validity_bit0, int64_t n,
                         int32_t* ids, int64_t* dict_rows, int32_t dict_cap, int32_t* n_values, int32_t threads);

/* Column of 64-bit keys -> ids[n] in first-occurrence order and dict_rows[*n_values] (one row that holds each key).  The front
 * door runs it over the object POINTERS of a pandas object-dtype column: rows that share a string object (the CSV parser interns
 * them per column) get one dictionary entry without the string being looked at, and only the few distinct objects are converted.
 * IVJ_ECAPACITY: more than dict_cap (or 4096) distinct keys -- the caller falls back to the ordinary conversion. */
int ivj_host_encode_keys64(const uint64_t* keys, int64_t n, int32_t* ids, int64_t* dict_rows, int32_t dict_cap, int32_t* n_values, int32_t threads);

/* Dictionary indices (idx_bytes = 1, 2, 4, 8, signed; negative = null) -> out[i] = remap[idx[i]] (-1 for a null), and seen[v] = 1
 * for every dictionary entry some row refers to (seen: remap_len bytes, OR-ed into). */
int ivj_host_remap_i32(const void* idx, int32_t idx_bytes, int64_t n, const int32_t* remap, int64_t remap_len, int32_t* out, uint8_t* seen,
                       int32_t threads);

/* dst[i] = src[idx[i]] for 4- or 8-byte values (0 for a negative index): the non-key columns of the joined rows. */
int ivj_host_take(const void* src, int32_t elem_bytes, int64_t n_src, const int32_t* idx, int64_t n, void* dst, int32_t threads);

/* dst[idx[i]] = src[i] for rows of row_bytes bytes: the mirror of ivj_host_take -- a shard's per-probe results (counts, nearest rows and
 * distances) back to their GLOBAL probe rows when one process drives several devices (threaded; the indices of one call are distinct;
 * an index outside [0, n_dst) is refused before anything is written).  remap (rows of int32 values only, NULL: none): the value stored is
 * remap[src value] for a value in [0, remap_len) and -1 otherwise -- a shard's local build rows become global rows on the way. */
int ivj_host_scatter(const void* src, int32_t row_bytes, int64_t n, const int32_t* idx, int64_t n_dst, void* dst, const int32_t* remap, int64_t remap_len,
                     int32_t threads);

/* Contig sharding of one side for `world` ranks ("host buckets both sides by contig id"; the reference's
 * partitioner cuts by row count, src/scan.rs:233-277): owner[c] = rank of contig c (n_contigs entries), a row whose contig lies
 * outside [0, n_contigs) belongs to no rank.  One counting pass fills counts[world]; with output columns (arrays of `world`
 * pointers, each sized from a first counts-only call with all four NULL; any of the four may be NULL) one placing pass writes
 * every rank's rows -- contig / start / end and the GLOBAL row (what ivj_side.row_id carries) -- in input order. */
int ivj_host_shard(const int32_t* contig, const int32_t* start, const int32_t* end, int64_t n, const int32_t* owner, int32_t n_contigs, int32_t world,
                   int64_t* counts, int32_t* const* out_contig, int32_t* const* out_start, int32_t* const* out_end, int32_t* const* out_row, int32_t threads);
/* rows per contig (the weights of the LPT contig -> rank assignment): hist[n_contigs] */
int ivj_host_contig_hist(const int32_t* contig, int64_t n, int32_t n_contigs, int64_t* hist, int32_t threads);

/* int32 -> int64: key columns materialised in HBM back to the dtype of the caller's frame. */
int ivj_host_widen_i32(const int32_t* src, int64_t n, int64_t* dst, int32_t threads);

#ifdef __cplusplus
}
#endif
#endif /* IVJOIN_H */
