// host_sortscan.hip.h -- drivers of the sort-scan family: cluster sweep, coverage, union, subtract
// Part of the single translation unit ivjoin.hip (included there, in this order); not a stand-alone header.
#pragma once

namespace {

// ---- sort-scan family (sortscan.hip.h) ------------------------------------------------------------
struct Clusters {                 // arena-backed (valid until the next arena_reserve on this context)
    int64_t n = 0;                // number of clusters
    uint32_t* cid1 = nullptr;     // per sorted position: 1-based cluster id
    int32_t *m_contig = nullptr, *m_start = nullptr, *m_end = nullptr, *m_first = nullptr;
};

int cluster_core(ivj_ctx* ctx, ivj_index* ix, bool strict, long long min_dist, size_t extra_bytes, Clusters& cl) {
    const int64_t n = ix->n;
    cl = Clusters();
    if (n == 0) return IVJ_OK;
    const size_t col = align_up((size_t)(n + 1) * 4);
    IVJ_TRY(arena_reserve(ctx, 6 * col + align_up((size_t)(scan_num_tiles(n) + 1) * 4) + extra_bytes + 4096));
    uint32_t* flags = arena_take<uint32_t>(ctx, n + 1);
    cl.cid1 = arena_take<uint32_t>(ctx, n + 1);
    cl.m_contig = arena_take<int32_t>(ctx, n + 1);
    cl.m_start = arena_take<int32_t>(ctx, n + 1);
    cl.m_end = arena_take<int32_t>(ctx, n + 1);
    cl.m_first = arena_take<int32_t>(ctx, n + 1);
    uint32_t* partials = arena_take<uint32_t>(ctx, scan_num_tiles(n) + 1);
    if (strict) LAUNCH(ctx, "cluster_flags", (k_cluster_flags<true>), grid1d(n, 256), 256, (const int32_t*)ix->b_start, (const int2*)ix->ep, (const int32_t*)ix->b_contig, n, min_dist, flags);
    else LAUNCH(ctx, "cluster_flags", (k_cluster_flags<false>), grid1d(n, 256), 256, (const int32_t*)ix->b_start, (const int2*)ix->ep, (const int32_t*)ix->b_contig, n, min_dist, flags);
    device_scan<uint32_t, SumOp, true>(ctx, "cluster_scan", flags, cl.cid1, n, 0u, partials, (uint32_t*)nullptr);
    LAUNCH(ctx, "cluster_bounds", k_cluster_bounds, grid1d(n, 256), 256, (const uint32_t*)flags, (const uint32_t*)cl.cid1, (const int32_t*)ix->b_start,
           (const int2*)ix->ep, (const int32_t*)ix->b_contig, n, ix->n_contigs, cl.m_contig, cl.m_start, cl.m_end, cl.m_first);
    uint32_t last = 0;
    HIP_TRY(hipMemcpyAsync(&last, cl.cid1 + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipGetLastError());
    cl.n = (int64_t)last;
    return IVJ_OK;
}

// pb.coverage through the union grid (sortscan.hip.h, k_coverage_grid): cluster sweep -> clipped lengths + prefix -> grid
// metadata + one 16-byte record per bin -> ONE kernel over the probes in their own order (no bucketing, no inverse
// permutation, no start table).  partition_mode 1 keeps the round-1 path (bucketed probes, table lookups) for A/B runs.
int coverage_grid(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int64_t* cov) {
    const int64_t n = probe->n;
    const bool strict = opts->filter_op == IVJ_FILTER_STRICT;
    const int nc = ix->n_contigs;
    Clusters cl;
    const int64_t max_slots = 2 * (ix->n + 1) + 2 * (int64_t)nc + 16;
    const size_t extra = 2 * align_up((size_t)(ix->n + 2) * 8) + align_up((size_t)(scan_num_tiles(ix->n + 1) + 1) * 8) +
                         align_up((size_t)(nc + 1) * 32) + align_up((size_t)max_slots * 16);
    IVJ_TRY(cluster_core(ctx, ix, strict, 0, extra, cl));
    long long* len = arena_take<long long>(ctx, ix->n + 2);
    long long* pl = arena_take<long long>(ctx, ix->n + 2);
    long long* partials = arena_take<long long>(ctx, scan_num_tiles(ix->n + 1) + 1);
    int4* cm = arena_take<int4>(ctx, 2 * (size_t)(nc + 1));
    int4* rec = arena_take<int4>(ctx, (size_t)max_slots);
    if (strict) LAUNCH(ctx, "merged_lengths", (k_merged_lengths<true>), grid1d(cl.n, 256), 256, (const int32_t*)cl.m_start, (const int32_t*)cl.m_end, cl.n, len);
    else LAUNCH(ctx, "merged_lengths", (k_merged_lengths<false>), grid1d(cl.n, 256), 256, (const int32_t*)cl.m_start, (const int32_t*)cl.m_end, cl.n, len);
    HIP_TRY(hipMemsetAsync(len + cl.n, 0, 8, ctx->stream));      // one padding element: pl[n_clusters] = total
    device_scan<long long, SumOp, false>(ctx, "merged_scan", len, pl, cl.n + 1, 0ll, partials, (long long*)nullptr);
    const int64_t n_slots = 2 * cl.n + 2 * (int64_t)nc + 2;
    if (strict) {
        LAUNCH(ctx, "coverage_meta", (k_cov_meta<true>), grid1d(nc, 256), 256, (const int32_t*)ix->seg, (const uint32_t*)cl.cid1, (const int32_t*)cl.m_start,
               (const int32_t*)cl.m_end, (const long long*)pl, nc, cm);
        LAUNCH(ctx, "coverage_records", (k_cov_records<true>), grid1d(n_slots, 256), 256, (const int4*)cm, nc, n_slots, (const int32_t*)cl.m_start,
               (const int32_t*)cl.m_end, (const long long*)pl, rec);
    } else {
        LAUNCH(ctx, "coverage_meta", (k_cov_meta<false>), grid1d(nc, 256), 256, (const int32_t*)ix->seg, (const uint32_t*)cl.cid1, (const int32_t*)cl.m_start,
               (const int32_t*)cl.m_end, (const long long*)pl, nc, cm);
        LAUNCH(ctx, "coverage_records", (k_cov_records<false>), grid1d(n_slots, 256), 256, (const int4*)cm, nc, n_slots, (const int32_t*)cl.m_start,
               (const int32_t*)cl.m_end, (const long long*)pl, rec);
    }
    const CovMeta g{cm, rec};
    const int32_t *qc = probe->contig, *qs = probe->start, *qe = probe->end;
    const bool vec = aligned16(qc) && aligned16(qs) && aligned16(qe);
    const int64_t per = (int64_t)PROBE_THREADS * COV2_ITEMS * COV2_TILES_PER_WG;
    const unsigned grid = (unsigned)((n + per - 1) / per);
    const bool lm = nc <= CM_LDS;
#define IVJ_COV_LAUNCH(S, L) LAUNCH(ctx, "coverage", (k_coverage_grid<S, L>), grid, PROBE_THREADS, g, nc, (const int32_t*)cl.m_start, \
                                    (const int32_t*)cl.m_end, (const long long*)pl, qc, qs, qe, n, vec, (long long*)cov)
    if (strict) { if (lm) IVJ_COV_LAUNCH(true, true); else IVJ_COV_LAUNCH(true, false); }
    else { if (lm) IVJ_COV_LAUNCH(false, true); else IVJ_COV_LAUNCH(false, false); }
#undef IVJ_COV_LAUNCH
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

int coverage_core(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int64_t* cov) {
    if (!ix->has_tables) IVJ_TRY(need_tables(ctx, ix));          // refuses a sweep-only index
    const int64_t n = probe->n;
    if (n == 0) return IVJ_OK;
    if (ix->n == 0) { HIP_TRY(hipMemsetAsync(cov, 0, (size_t)n * 8, ctx->stream)); return IVJ_OK; }
    if (opts->partition_mode != 1 && ix->n_contigs > 0) return coverage_grid(ctx, ix, probe, opts, cov);
    IVJ_TRY(need_tables(ctx, ix));
    const bool strict = opts->filter_op == IVJ_FILTER_STRICT;
    const bool bucketed = want_partition(ix, n, opts) && !probe->row_id;
    if (bucketed) {                                          // before cluster_core: the partition uses the arena too
        ivj_side plain = *probe;
        IVJ_TRY(ensure_ov(ctx, n, 1));
        ctx->ov_n = -1;
        ivj_opts popts = *opts; popts.partition_mode = 1;
        IVJ_TRY(partition_probes(ctx, ix, &plain, &popts));
    }
    Clusters cl;
    const size_t extra = 2 * align_up((size_t)(ix->n + 2) * 8) + align_up((size_t)(scan_num_tiles(ix->n + 1) + 1) * 8) +
                         (bucketed ? align_up((size_t)n * 8) : 0);
    IVJ_TRY(cluster_core(ctx, ix, strict, 0, extra, cl));
    long long* len = arena_take<long long>(ctx, ix->n + 2);
    long long* pl = arena_take<long long>(ctx, ix->n + 2);
    long long* partials = arena_take<long long>(ctx, scan_num_tiles(ix->n + 1) + 1);
    if (strict) LAUNCH(ctx, "merged_lengths", (k_merged_lengths<true>), grid1d(cl.n, 256), 256, (const int32_t*)cl.m_start, (const int32_t*)cl.m_end, cl.n, len);
    else LAUNCH(ctx, "merged_lengths", (k_merged_lengths<false>), grid1d(cl.n, 256), 256, (const int32_t*)cl.m_start, (const int32_t*)cl.m_end, cl.n, len);
    HIP_TRY(hipMemsetAsync(len + cl.n, 0, 8, ctx->stream));      // one padding element: pl[n_clusters] = total
    device_scan<long long, SumOp, false>(ctx, "merged_scan", len, pl, cl.n + 1, 0ll, partials, (long long*)nullptr);
    IndexView v = view_of(ix);
    // large probe sides: bucket them by genomic position first (the table / cluster gathers then stay in L2);
    // the kernel writes each result to the probe's original row
    const int32_t *qc = probe->contig, *qs = probe->start, *qe = probe->end, *qrow = nullptr;
    if (bucketed) { qc = ctx->pt_c; qs = ctx->pt_s; qe = ctx->pt_e; qrow = ctx->pt_row; }
    const bool vec = aligned16(qc) && aligned16(qs) && aligned16(qe);
    const int64_t per = (int64_t)PROBE_THREADS * COV_ITEMS;
    long long* o_cov = bucketed ? arena_take<long long>(ctx, n) : (long long*)cov;    // bucket order, un-permuted below
    (void)qrow;
    if (strict) LAUNCH(ctx, "coverage", (k_coverage<true>), 8 * (((n + per - 1) / per + 7) / 8), PROBE_THREADS, v, (const uint32_t*)cl.cid1,
                       (const int32_t*)cl.m_start, (const int32_t*)cl.m_end, (const long long*)pl, qc, qs, qe, (const int32_t*)nullptr, n, vec, o_cov);
    else LAUNCH(ctx, "coverage", (k_coverage<false>), 8 * (((n + per - 1) / per + 7) / 8), PROBE_THREADS, v, (const uint32_t*)cl.cid1,
                (const int32_t*)cl.m_start, (const int32_t*)cl.m_end, (const long long*)pl, qc, qs, qe, (const int32_t*)nullptr, n, vec, o_cov);
    if (bucketed) {
        UnpermuteCols uc{{o_cov, nullptr, nullptr}, {cov, nullptr, nullptr}, {8, 0, 0}, 1, nullptr};
        IVJ_TRY(unpermute(ctx, n, uc));
    }
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

// union of the index's intervals as compacted half-open int64 ranges + everything k_subtract_* needs
struct UnionView {
    Clusters cl;
    uint32_t *keep = nullptr, *newidx = nullptr;
    long long *u_start = nullptr, *u_end = nullptr;
};

int union_core(ivj_ctx* ctx, ivj_index* ix, bool strict, size_t extra_bytes, UnionView& u) {
    const int64_t n = ix->n;
    const size_t mine = 2 * align_up((size_t)(n + 2) * 4) + 2 * align_up((size_t)(n + 2) * 8) + align_up((size_t)(scan_num_tiles(n + 1) + 1) * 4);
    IVJ_TRY(cluster_core(ctx, ix, strict, 1, mine + extra_bytes, u.cl));
    if (n == 0) return IVJ_OK;
    uint32_t* keep = u.keep = arena_take<uint32_t>(ctx, n + 2);
    u.newidx = arena_take<uint32_t>(ctx, n + 2);
    u.u_start = arena_take<long long>(ctx, n + 2);
    u.u_end = arena_take<long long>(ctx, n + 2);
    uint32_t* partials = arena_take<uint32_t>(ctx, scan_num_tiles(n + 1) + 1);
    const int64_t ncl = u.cl.n;
    if (strict) LAUNCH(ctx, "union_flags", (k_union_flags<true>), grid1d(ncl, 256), 256, (const int32_t*)u.cl.m_start, (const int32_t*)u.cl.m_end, ncl, keep);
    else LAUNCH(ctx, "union_flags", (k_union_flags<false>), grid1d(ncl, 256), 256, (const int32_t*)u.cl.m_start, (const int32_t*)u.cl.m_end, ncl, keep);
    HIP_TRY(hipMemsetAsync(keep + ncl, 0, 4, ctx->stream));
    device_scan<uint32_t, SumOp, false>(ctx, "union_scan", keep, u.newidx, ncl + 1, 0u, partials, (uint32_t*)nullptr);
    if (strict) LAUNCH(ctx, "union_compact", (k_union_compact<true>), grid1d(ncl, 256), 256, (const int32_t*)u.cl.m_start, (const int32_t*)u.cl.m_end,
                       (const uint32_t*)keep, (const uint32_t*)u.newidx, ncl, u.u_start, u.u_end);
    else LAUNCH(ctx, "union_compact", (k_union_compact<false>), grid1d(ncl, 256), 256, (const int32_t*)u.cl.m_start, (const int32_t*)u.cl.m_end,
                (const uint32_t*)keep, (const uint32_t*)u.newidx, ncl, u.u_start, u.u_end);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

// subtract / complement through the union grid (sortscan.hip.h, k_subtract_grid): union -> grid metadata + one 16-byte
// record per bin -> count pass, scan, fill pass over the left rows in their own order (no bucketing, no start table).
int subtract_grid(ivj_ctx* ctx, ivj_index* ix, const ivj_side* left, const ivj_opts* opts, int64_t capacity, int32_t** o_row,
                  int32_t** o_start, int32_t** o_end, DevBuf* own, int64_t* n_pieces) {
    const int64_t n = left->n;
    const bool strict = opts->filter_op == IVJ_FILTER_STRICT;
    const int nc = ix->n_contigs;
    const int64_t max_slots = 2 * (ix->n + 1) + 2 * (int64_t)nc + 16;
    const size_t extra = 2 * align_up((size_t)(n + 1) * 8) + align_up((size_t)(scan_num_tiles(n) + 2) * 8) + 256 +
                         align_up((size_t)(nc + 1) * 32) + align_up((size_t)max_slots * 16);
    UnionView u;
    IVJ_TRY(union_core(ctx, ix, strict, extra, u));
    long long* cnt = arena_take<long long>(ctx, n + 1);
    long long* off = arena_take<long long>(ctx, n + 1);
    long long* partials = arena_take<long long>(ctx, scan_num_tiles(n) + 2);
    int4* cm = arena_take<int4>(ctx, 2 * (size_t)(nc + 1));
    int4* rec = arena_take<int4>(ctx, (size_t)max_slots);
    const int64_t n_slots = 2 * u.cl.n + 2 * (int64_t)nc + 2;
    LAUNCH(ctx, "subtract_meta", k_sub_meta, grid1d(nc, 256), 256, (const int32_t*)ix->seg, (const uint32_t*)u.cl.cid1, (const uint32_t*)u.newidx,
           (const long long*)u.u_start, (const long long*)u.u_end, nc, cm);
    LAUNCH(ctx, "subtract_records", k_sub_records, grid1d(n_slots, 256), 256, (const int4*)cm, nc, n_slots, (const long long*)u.u_start,
           (const long long*)u.u_end, rec);
    const SubGrid g{cm, rec};
    if (strict) LAUNCH(ctx, "subtract_count", (k_subtract_grid<true, 0>), grid1d(n, PROBE_THREADS), PROBE_THREADS, g, nc, (const long long*)u.u_start, (const long long*)u.u_end,
                       left->contig, left->start, left->end, left->row_id, n, cnt, (const long long*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
    else LAUNCH(ctx, "subtract_count", (k_subtract_grid<false, 0>), grid1d(n, PROBE_THREADS), PROBE_THREADS, g, nc, (const long long*)u.u_start, (const long long*)u.u_end,
                left->contig, left->start, left->end, left->row_id, n, cnt, (const long long*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
    long long* total_dev = partials + scan_num_tiles(n) + 1;
    device_scan<long long, SumOp, false>(ctx, "subtract_scan", cnt, off, n, 0ll, partials, total_dev);
    HIP_TRY(hipMemcpyAsync(ctx->h_total, total_dev, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const int64_t total = ctx->h_total[0];
    *n_pieces = total;
    if (total == 0) return IVJ_OK;
    if (capacity < 0) {
        const size_t col = align_up((size_t)total * 4);
        hipError_t e = hipMalloc(&own->p, 3 * col);
        if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string("hipMalloc(pieces): ") + hipGetErrorString(e));
        *o_row = (int32_t*)own->p; *o_start = (int32_t*)((char*)own->p + col); *o_end = (int32_t*)((char*)own->p + 2 * col);
    } else if (total > capacity) {
        return fail(IVJ_ECAPACITY, "output capacity " + std::to_string(capacity) + " < " + std::to_string(total) + " pieces");
    } else if (!*o_row || !*o_start || !*o_end) {
        return fail(IVJ_EINVAL, "subtract output buffers are NULL");
    }
    if (strict) LAUNCH(ctx, "subtract_fill", (k_subtract_grid<true, 1>), grid1d(n, PROBE_THREADS), PROBE_THREADS, g, nc, (const long long*)u.u_start, (const long long*)u.u_end,
                       left->contig, left->start, left->end, left->row_id, n, (long long*)nullptr, (const long long*)off, *o_row, *o_start, *o_end);
    else LAUNCH(ctx, "subtract_fill", (k_subtract_grid<false, 1>), grid1d(n, PROBE_THREADS), PROBE_THREADS, g, nc, (const long long*)u.u_start, (const long long*)u.u_end,
                left->contig, left->start, left->end, left->row_id, n, (long long*)nullptr, (const long long*)off, *o_row, *o_start, *o_end);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

// left minus the union of the index.  capacity < 0: library-allocated device outputs (host path), otherwise the
// caller's buffers; *n_pieces always receives the total.
int subtract_core(ivj_ctx* ctx, ivj_index* ix, const ivj_side* left, const ivj_opts* opts, int64_t capacity, int32_t** o_row,
                  int32_t** o_start, int32_t** o_end, DevBuf* own, int64_t* n_pieces) {
    if (!ix->has_tables) IVJ_TRY(need_tables(ctx, ix));          // refuses a sweep-only index
    const int64_t n = left->n;
    *n_pieces = 0;
    if (n == 0) return IVJ_OK;
    if (opts->partition_mode != 1 && ix->n > 0 && ix->n_contigs > 0) return subtract_grid(ctx, ix, left, opts, capacity, o_row, o_start, o_end, own, n_pieces);
    IVJ_TRY(need_tables(ctx, ix));
    const bool strict = opts->filter_op == IVJ_FILTER_STRICT;
    const bool bucketed = want_partition(ix, n, opts) && ix->n > 0;
    if (bucketed) {                                          // before union_core: the partition uses the arena too
        ivj_side plain = *left;
        plain.row_id = nullptr;                              // pt_row = position in the caller's columns
        IVJ_TRY(ensure_ov(ctx, n, 1));
        ctx->ov_n = -1;
        ivj_opts popts = *opts; popts.partition_mode = 1;
        IVJ_TRY(partition_probes(ctx, ix, &plain, &popts));
    }
    const int32_t *lc = left->contig, *lst = left->start, *len_ = left->end, *lpos = nullptr;
    if (bucketed) { lc = ctx->pt_c; lst = ctx->pt_s; len_ = ctx->pt_e; lpos = ctx->pt_row; }
    const size_t extra = 2 * align_up((size_t)(n + 1) * 8) + align_up((size_t)(scan_num_tiles(n) + 2) * 8) + 256;
    UnionView u;
    IVJ_TRY(union_core(ctx, ix, strict, extra, u));
    if (ix->n == 0) {
        // nothing to subtract: union_core took nothing from the arena, reserve the per-row arrays here
        IVJ_TRY(arena_reserve(ctx, extra + 4096));
    }
    long long* cnt = arena_take<long long>(ctx, n + 1);
    long long* off = arena_take<long long>(ctx, n + 1);
    long long* partials = arena_take<long long>(ctx, scan_num_tiles(n) + 2);
    IndexView v = view_of(ix);
    // an empty index has zeroed segment offsets: every row then keeps its one piece
    if (strict) LAUNCH(ctx, "subtract_count", (k_subtract_count<true>), grid1d(n, PROBE_THREADS), PROBE_THREADS, v, (const uint32_t*)u.cl.cid1, (const uint32_t*)u.keep, (const uint32_t*)u.newidx,
                       (const long long*)u.u_start, (const long long*)u.u_end, lc, lst, len_, lpos, n, cnt);
    else LAUNCH(ctx, "subtract_count", (k_subtract_count<false>), grid1d(n, PROBE_THREADS), PROBE_THREADS, v, (const uint32_t*)u.cl.cid1, (const uint32_t*)u.keep, (const uint32_t*)u.newidx,
                (const long long*)u.u_start, (const long long*)u.u_end, lc, lst, len_, lpos, n, cnt);
    long long* total_dev = partials + scan_num_tiles(n) + 1;
    device_scan<long long, SumOp, false>(ctx, "subtract_scan", cnt, off, n, 0ll, partials, total_dev);
    HIP_TRY(hipMemcpyAsync(ctx->h_total, total_dev, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const int64_t total = ctx->h_total[0];
    *n_pieces = total;
    if (total == 0) return IVJ_OK;
    if (capacity < 0) {
        const size_t col = align_up((size_t)total * 4);
        hipError_t e = hipMalloc(&own->p, 3 * col);
        if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string("hipMalloc(pieces): ") + hipGetErrorString(e));
        *o_row = (int32_t*)own->p; *o_start = (int32_t*)((char*)own->p + col); *o_end = (int32_t*)((char*)own->p + 2 * col);
    } else if (total > capacity) {
        return fail(IVJ_ECAPACITY, "output capacity " + std::to_string(capacity) + " < " + std::to_string(total) + " pieces");
    } else if (!*o_row || !*o_start || !*o_end) {
        return fail(IVJ_EINVAL, "subtract output buffers are NULL");
    }
    if (strict) LAUNCH(ctx, "subtract_fill", (k_subtract_fill<true>), grid1d(n, PROBE_THREADS), PROBE_THREADS, v, (const uint32_t*)u.cl.cid1, (const uint32_t*)u.keep, (const uint32_t*)u.newidx,
                       (const long long*)u.u_start, (const long long*)u.u_end, lc, lst, len_, lpos, left->row_id, n,
                       (const long long*)off, *o_row, *o_start, *o_end);
    else LAUNCH(ctx, "subtract_fill", (k_subtract_fill<false>), grid1d(n, PROBE_THREADS), PROBE_THREADS, v, (const uint32_t*)u.cl.cid1, (const uint32_t*)u.keep, (const uint32_t*)u.newidx,
                (const long long*)u.u_start, (const long long*)u.u_end, lc, lst, len_, lpos, left->row_id, n,
                (const long long*)off, *o_row, *o_start, *o_end);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

}  // namespace
