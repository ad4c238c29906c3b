// host_join.hip.h -- probe bucketing and the overlap / count_overlaps / nearest drivers, host <-> device staging of the host-buffer entry points
// Part of the single translation unit ivjoin.hip (included there, in this order); not a stand-alone header.
#pragma once

namespace {

int ensure_ov(ivj_ctx* ctx, int64_t n, int with_part) {      // 0: none, 1: one permuted column set
    const int64_t tiles = (n + PROBE_TILE - 1) / PROBE_TILE;
    const size_t col = align_up((size_t)n * 4);
    const size_t ptiles = (size_t)((n + PART_TILE - 1) / PART_TILE);
    const size_t off_bytes = with_part ? align_up((size_t)PART_BUCKETS * ptiles * 4 + 64) : 0;
    const size_t need = (size_t)(2 + 4 * with_part) * col + off_bytes + align_up((size_t)(tiles + 2) * 8) +
                        align_up((size_t)(scan_num_tiles(tiles) + 2) * 8) + align_up((PART_BUCKETS + 1) * 4) + 1024;
    if (need > ctx->ov_cap) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->ov_buf) HIP_TRY(hipFree(ctx->ov_buf));
        ctx->ov_buf = nullptr; ctx->ov_cap = 0;
        size_t want = align_up(need + need / 8, 1 << 20);
        hipError_t e = hipMalloc((void**)&ctx->ov_buf, want);
        if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string("overlap state hipMalloc: ") + hipGetErrorString(e));
        ctx->ov_cap = want;
    }
    char* p = ctx->ov_buf;
    ctx->ov_hi = (int32_t*)p; p += col;
    ctx->ov_cnt = (int32_t*)p; p += col;
    if (with_part) {
        ctx->pt_c = (int32_t*)p; p += col;
        ctx->pt_s = (int32_t*)p; p += col;
        ctx->pt_e = (int32_t*)p; p += col;
        ctx->pt_row = (int32_t*)p; p += col;
        ctx->pt_off = (uint32_t*)p; p += off_bytes;            // scanned per-tile histogram of the partition (k_unpermute reads it)
    }
    ctx->pt_bstart = (uint32_t*)p; p += align_up((PART_BUCKETS + 1) * 4);
    ctx->ov_tile = (long long*)p;
    return IVJ_OK;
}

// Probe bucketing pays once the index no longer fits the L2s and there are enough probes to
// amortise the two extra passes.  opts->partition_mode: 0 auto, 1 always, 2 never.
bool want_partition(const ivj_index* ix, int64_t n_probe, const ivj_opts* opts) {
    if (opts->partition_mode == 1 || opts->partition_mode >= 5) return true;
    if (opts->partition_mode == 2) return false;
    return n_probe >= (4ll << 20) && ix->n >= (256ll << 10);
}

// one stable 256-way pass: src columns -> dst columns
int partition_pass(ivj_ctx* ctx, ivj_index* ix, bool strict, const int32_t* sc, const int32_t* ss, const int32_t* se,
                   const int32_t* srow, int64_t n, int bshift, int32_t* dc, int32_t* ds, int32_t* de, int32_t* drow) {
    const int ntiles = (int)((n + PART_TILE - 1) / PART_TILE);
    const int grid = 8 * ((ntiles + 7) / 8);
    const size_t hist = (size_t)PART_BUCKETS * (size_t)ntiles;
    uint32_t* blk = ctx->pt_off;                               // outlives the arena: the inverse permutation reads it
    ctx->pt_ntiles = ntiles;
    if (!ctx->part_attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_part_scatter<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PART_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_part_scatter<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PART_LDS_BYTES));
        ctx->part_attr_set = true;
    }
    IndexView v = view_of(ix);
    const bool hvec = aligned16(sc) && aligned16(se);
    if (strict) LAUNCH(ctx, "part_hist", (k_part_hist<true>), grid, PART_THREADS, v, sc, se, n, bshift, blk, ntiles, hvec);
    else LAUNCH(ctx, "part_hist", (k_part_hist<false>), grid, PART_THREADS, v, sc, se, n, bshift, blk, ntiles, hvec);
    IVJ_TRY((lb_scan_u32<SumOp, true>(ctx, "part_scan", blk, (int64_t)hist, 0u)));
    // bucket b starts at blk[b * ntiles] (bucket-major scan); kept for the inverse permutation (k_unpermute)
    HIP_TRY(hipMemcpy2DAsync(ctx->pt_bstart, 4, blk, (size_t)ntiles * 4, 4, PART_BUCKETS, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(ctx->pt_bstart + PART_BUCKETS), (int)n, 1, ctx->stream));
    t_begin(ctx, "part_scatter");
    if (strict)
        hipLaunchKernelGGL((k_part_scatter<true>), dim3(grid), dim3(PART_THREADS), PART_LDS_BYTES, ctx->stream, v, sc, ss, se, srow, n, bshift,
                           (const uint32_t*)blk, ntiles, dc, ds, de, drow);
    else
        hipLaunchKernelGGL((k_part_scatter<false>), dim3(grid), dim3(PART_THREADS), PART_LDS_BYTES, ctx->stream, v, sc, ss, se, srow, n, bshift,
                           (const uint32_t*)blk, ntiles, dc, ds, de, drow);
    t_end(ctx);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

int partition_probes(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts) {
    const int64_t n = probe->n;
    const bool strict = opts->filter_op == IVJ_FILTER_STRICT;
    int bshift = 0;
    while ((ix->bins_len >> bshift) > (int64_t)(PART_BUCKETS - 3)) ++bshift;
    return partition_pass(ctx, ix, strict, probe->contig, probe->start, probe->end, probe->row_id, n, bshift,
                          ctx->pt_c, ctx->pt_s, ctx->pt_e, ctx->pt_row);
}

int overlap_count(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int64_t* n_pairs) {
    const int64_t n = probe->n;
    ctx->ov_n = -1;
    ctx->ov_slice = false;
    if (n == 0 || ix->n == 0) {
        ctx->ov_n = n; ctx->ov_total = 0; ctx->ov_probe_start = probe->start; ctx->ov_probe_contig = probe->contig; ctx->ov_probe_end = probe->end; ctx->ov_ix = ix; ctx->ov_filter = opts->filter_op;
        *n_pairs = 0;
        return IVJ_OK;
    }
    SliceGeom sg;
    if (want_slices(ix, n, opts, sg, false)) {
        int64_t total = 0;
        ctx->ov_cs = cs_wanted(ctx, ix, opts);
        if (ctx->ov_cs) IVJ_TRY(cs_overlap_count(ctx, ix, probe, opts, &total));
        else IVJ_TRY(slice_overlap_count(ctx, ix, probe, opts, sg, &total));
        ctx->ov_slice = true;
        ctx->ov_total = total;
        ctx->ov_n = n; ctx->ov_probe_start = probe->start; ctx->ov_probe_contig = probe->contig; ctx->ov_probe_end = probe->end; ctx->ov_ix = ix; ctx->ov_filter = opts->filter_op;
        *n_pairs = total;
        return IVJ_OK;
    }
    IVJ_TRY(need_tables(ctx, ix));
    IVJ_TRY(ensure_hier(ctx, ix));                     // windows longer than the mask walk the block maxima
    const bool part = want_partition(ix, n, opts);
    IVJ_TRY(ensure_ov(ctx, n, part ? 1 : 0));
    ctx->ov_part = part;
    const int64_t tiles = (n + PROBE_TILE - 1) / PROBE_TILE;
    long long* tile = ctx->ov_tile;                       // tiles + 1
    long long* partials = tile + align_up((size_t)(tiles + 2) * 8) / 8;
    const int32_t *qc = probe->contig, *qs = probe->start, *qe = probe->end;
    if (part) {
        IVJ_TRY(partition_probes(ctx, ix, probe, opts));
        qc = ctx->pt_c; qs = ctx->pt_s; qe = ctx->pt_e;
    }
    const bool vec = aligned16(qc) && aligned16(qs) && aligned16(qe);
    IndexView v = view_of(ix);
    if (opts->filter_op == IVJ_FILTER_STRICT)
        LAUNCH(ctx, "overlap_count", (k_overlap_count<true>), 8 * ((tiles + 7) / 8), PROBE_THREADS, v, qc, qs, qe, n, vec,
               ctx->ov_hi, ctx->ov_cnt, tile);
    else
        LAUNCH(ctx, "overlap_count", (k_overlap_count<false>), 8 * ((tiles + 7) / 8), PROBE_THREADS, v, qc, qs, qe, n, vec,
               ctx->ov_hi, ctx->ov_cnt, tile);
    device_scan<long long, SumOp, false>(ctx, "tile_scan", tile, tile, tiles, 0ll, partials, tile + tiles);
    HIP_TRY(hipMemcpyAsync(ctx->h_total, tile + tiles, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipGetLastError());
    ctx->ov_total = *ctx->h_total;
    ctx->ov_n = n; ctx->ov_probe_start = probe->start; ctx->ov_probe_contig = probe->contig; ctx->ov_probe_end = probe->end; ctx->ov_ix = ix; ctx->ov_filter = opts->filter_op;
    *n_pairs = ctx->ov_total;
    return IVJ_OK;
}

int overlap_fused(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int32_t* out_p, int32_t* out_b,
                  int64_t capacity, int64_t* n_pairs);

int overlap_fill(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int32_t* out_p, int32_t* out_b,
                 int64_t capacity) {
    // (the dense reroute below re-reads all three probe columns, so all three are part of the hand-over's identity)
    if (ctx->ov_n != probe->n || ctx->ov_probe_start != probe->start || ctx->ov_probe_contig != probe->contig || ctx->ov_probe_end != probe->end ||
        ctx->ov_ix != ix || ctx->ov_filter != opts->filter_op)
        return fail(IVJ_ESTATE, "ivj_overlap_fill_dev must follow ivj_overlap_count_dev with the same index, probe and filter_op");
    if (capacity < ctx->ov_total) return fail(IVJ_ECAPACITY, "output capacity " + std::to_string(capacity) + " < " + std::to_string(ctx->ov_total) + " pairs");
    if (ctx->ov_total == 0) return IVJ_OK;
    if (!out_p || !out_b) return fail(IVJ_EINVAL, "output buffers are NULL");
    if (ctx->ov_slice && opts->partition_mode == 0 && !opts->deterministic && !ctx->sl_env_stable && ix->n_contigs > 0 && ctx->ov_total >= 16 * probe->n) {
        // DENSE result (>= 16 pairs per probe row, now known exactly): the slice FILL spends its time in the long windows -- 44 ms
        // for the 3.7 G pairs of the dense 100 M x 5 M variant -- where the flat kernel tests every candidate with its own lane
        // (12.5 ms + its tables).  The count is exact, so the fused flat pass runs with capacity = the count (round 5; a trace of
        // the dense bench showed the warm-up's count -> fill pair at 50 ms next to 15-ms fused steps: what pb.overlap ran).
        // The flat kernel orders its tiles with an atomic cursor: not for opts.deterministic / IVJ_SLICE_STABLE=1 callers (they keep the slice FILL).
        const int64_t want = ctx->ov_total;
        int64_t got = 0;
        IVJ_TRY(overlap_fused(ctx, ix, probe, opts, out_p, out_b, want, &got));
        if (got != want)
            return fail(IVJ_ESTATE, "the fill pass found " + std::to_string(got) + " pairs where the count pass found " + std::to_string(want) +
                                    ": the probe columns changed between ivj_overlap_count_dev and ivj_overlap_fill_dev");
        return IVJ_OK;
    }
    if (ctx->ov_slice) return ctx->ov_cs ? cs_overlap_fill(ctx, ix, opts, out_p, out_b) : slice_overlap_fill(ctx, ix, opts, out_p, out_b);
    const int64_t n = probe->n;
    const int64_t tiles = (n + PROBE_TILE - 1) / PROBE_TILE;
    const int32_t* qs = ctx->ov_part ? ctx->pt_s : probe->start;
    const int32_t* ids = ctx->ov_part ? ctx->pt_row : probe->row_id;
    const bool vec = aligned16(qs);
    IndexView v = view_of(ix);
    // dense results (>= 8 pairs per probe on average): windows shared out over all wavefronts
    const bool dense = ctx->ov_total >= 8 * n;
    const bool strict = opts->filter_op == IVJ_FILTER_STRICT;
    if (dense) {
        if (strict) LAUNCH(ctx, "overlap_fill_dense", (k_overlap_fill_dense<true, PROBE_ITEMS>), tiles, PROBE_THREADS, v, qs, n, vec, (const int32_t*)ctx->ov_hi,
                           (const int32_t*)ctx->ov_cnt, (const long long*)ctx->ov_tile, ids, out_p, out_b);
        else LAUNCH(ctx, "overlap_fill_dense", (k_overlap_fill_dense<false, PROBE_ITEMS>), tiles, PROBE_THREADS, v, qs, n, vec, (const int32_t*)ctx->ov_hi,
                    (const int32_t*)ctx->ov_cnt, (const long long*)ctx->ov_tile, ids, out_p, out_b);
    } else {
        if (strict) LAUNCH(ctx, "overlap_fill", (k_overlap_fill<true>), 8 * ((tiles + 7) / 8), PROBE_THREADS, v, qs, n, vec, (const int32_t*)ctx->ov_hi,
                           (const int32_t*)ctx->ov_cnt, (const long long*)ctx->ov_tile, ids, out_p, out_b);
        else LAUNCH(ctx, "overlap_fill", (k_overlap_fill<false>), 8 * ((tiles + 7) / 8), PROBE_THREADS, v, qs, n, vec, (const int32_t*)ctx->ov_hi,
                    (const int32_t*)ctx->ov_cnt, (const long long*)ctx->ov_tile, ids, out_p, out_b);
    }
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

// single pass: (bucketing +) fused count/fill into a caller buffer of known capacity
int overlap_fused(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int32_t* out_p, int32_t* out_b,
                  int64_t capacity, int64_t* n_pairs) {
    const int64_t n = probe->n;
    ctx->ov_n = -1;                                   // invalidates a pending count -> fill hand-over
    *n_pairs = 0;
    if (n == 0 || ix->n == 0) return IVJ_OK;
    {
        // sparse results of large inputs: LDS-resident index slices; dense ones (capacity says >= 16 pairs per probe) keep the flat kernel
        SliceGeom sg;
        const bool dense = opts->partition_mode == 0 && capacity >= 16 * n && ix->n_contigs > 0;
        if (!dense && want_slices(ix, n, opts, sg, true)) {
            // contig-aligned slices + 12-byte records (round 3) where the dictionary allows, the round-2 slice kernels otherwise
            if (cs_wanted(ctx, ix, opts)) return cs_overlap_fused(ctx, ix, probe, opts, out_p, out_b, capacity, n_pairs);
            return slice_overlap_fused(ctx, ix, probe, opts, sg, out_p, out_b, capacity, n_pairs);
        }
    }
    IVJ_TRY(need_tables(ctx, ix));
    IVJ_TRY(ensure_hier(ctx, ix));                     // windows longer than the mask walk the block maxima
    const bool part = want_partition(ix, n, opts);
    IVJ_TRY(ensure_ov(ctx, n, part ? 1 : 0));
    // dense results (the caller expects >= 16 pairs per probe; at ~8 the two kernels tie and the flat one still has
    // to fill its arrays and the end order): the flat kernel spreads every window over the whole
    // workgroup (1.8x the count + dense-fill pair on 37 pairs per probe); sparse ones keep the window-scan kernel
    bool flat = opts->partition_mode == 5 || (opts->partition_mode == 0 && capacity >= 16 * n && ix->n_contigs > 0);
    if (flat) IVJ_TRY(build_flat(ctx, ix));
    // dense tiles of the flat kernel get their match counts from the end order (two-rank formula) instead of a sweep
    const bool rank_counts = flat && capacity >= 16 * n;
    if (rank_counts) IVJ_TRY(build_end_table(ctx, ix));
    unsigned long long* state = (unsigned long long*)ctx->ov_tile;   // [0] cursor, [1] overflow, [2] flat kernel: some candidate range was too long
    const int32_t *qc = probe->contig, *qs = probe->start, *qe = probe->end, *ids = probe->row_id;
    if (part) {
        IVJ_TRY(partition_probes(ctx, ix, probe, opts));
        qc = ctx->pt_c; qs = ctx->pt_s; qe = ctx->pt_e; ids = ctx->pt_row;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int64_t tiles = flat ? (n + FLAT_TILE - 1) / FLAT_TILE : (n + PROBE_TILE - 1) / PROBE_TILE;
        HIP_TRY(hipMemsetAsync(state, 0, 24, ctx->stream));
        const bool vec = aligned16(qc) && aligned16(qs) && aligned16(qe);
        IndexView v = view_of(ix);
        if (flat) {
            if (opts->filter_op == IVJ_FILTER_STRICT)
                LAUNCH(ctx, "overlap_flat", (k_overlap_flat<true>), 8 * ((tiles + 7) / 8), FLAT_THREADS, v, qc, qs, qe, ids, n, vec, (long long)capacity, state, out_p, out_b, (int)rank_counts);
            else
                LAUNCH(ctx, "overlap_flat", (k_overlap_flat<false>), 8 * ((tiles + 7) / 8), FLAT_THREADS, v, qc, qs, qe, ids, n, vec, (long long)capacity, state, out_p, out_b, (int)rank_counts);
        } else if (opts->filter_op == IVJ_FILTER_STRICT)
            LAUNCH(ctx, "overlap_fused", (k_overlap_fused<true>), 8 * ((tiles + 7) / 8), PROBE_THREADS, v, qc, qs, qe, ids, n, vec, (long long)capacity, state, out_p, out_b);
        else
            LAUNCH(ctx, "overlap_fused", (k_overlap_fused<false>), 8 * ((tiles + 7) / 8), PROBE_THREADS, v, qc, qs, qe, ids, n, vec, (long long)capacity, state, out_p, out_b);
        HIP_TRY(hipMemcpyAsync(ctx->h_total, state, 24, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipGetLastError());
        // the flat kernel met a probe whose candidate range is a long sparse window (flat.hip.h FLAT_MAX_CAND): the window kernels redo the call
        if (flat && ctx->h_total[2] != 0) { flat = false; continue; }
        break;
    }
    *n_pairs = ctx->h_total[0];
    if (ctx->h_total[1] != 0)
        return fail(IVJ_ECAPACITY, "output capacity " + std::to_string(capacity) + " < " + std::to_string(ctx->h_total[0]) + " pairs");
    return IVJ_OK;
}

struct DevBuf {                   // owning device allocation of the host-buffer entry points
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

// per-probe results of a kernel that ran over the bucketed probes (pt_*) -> original row order
int unpermute(ivj_ctx* ctx, int64_t n, const UnpermuteCols& cols) {
    LAUNCH(ctx, "unpermute", k_unpermute, (n + UNP_TILE - 1) / UNP_TILE, UNP_THREADS, (const int32_t*)ctx->pt_row, (const uint32_t*)ctx->pt_bstart,
           (const uint32_t*)ctx->pt_off, ctx->pt_ntiles, n, cols);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

// fused join + key-column materialisation (k_overlap_fused_rows); same partitioning as overlap_fused
int overlap_fused_rows(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, const ivj_rows* rows, int64_t* n_pairs) {
    IVJ_TRY(need_tables(ctx, ix));
    IVJ_TRY(ensure_hier(ctx, ix));                     // windows longer than the mask walk the block maxima
    const int64_t n = probe->n;
    const int64_t capacity = rows->n_pairs;
    ctx->ov_n = -1;
    *n_pairs = 0;
    if (n == 0 || ix->n == 0) return IVJ_OK;
    IVJ_TRY(build_rec4(ctx, ix));
    const bool part = want_partition(ix, n, opts);
    IVJ_TRY(ensure_ov(ctx, n, part ? 1 : 0));
    const int64_t tiles = (n + PROBE_TILE - 1) / PROBE_TILE;
    unsigned long long* state = (unsigned long long*)ctx->ov_tile;
    const int32_t *qc = probe->contig, *qs = probe->start, *qe = probe->end, *ids = probe->row_id;
    if (part) {
        IVJ_TRY(partition_probes(ctx, ix, probe, opts));
        qc = ctx->pt_c; qs = ctx->pt_s; qe = ctx->pt_e; ids = ctx->pt_row;
    }
    HIP_TRY(hipMemsetAsync(state, 0, 16, ctx->stream));
    const bool vec = aligned16(qc) && aligned16(qs) && aligned16(qe) && aligned16(ids);
    IndexView v = view_of(ix);
    RowColumns cols{rows->probe_idx, rows->build_idx, rows->contig, rows->start_1, rows->end_1, rows->start_2, rows->end_2};
    if (opts->filter_op == IVJ_FILTER_STRICT)
        LAUNCH(ctx, "overlap_fused_rows", (k_overlap_fused_rows<true>), 8 * ((tiles + 7) / 8), PROBE_THREADS, v, qc, qs, qe, ids, n, vec, (long long)capacity, state, cols);
    else
        LAUNCH(ctx, "overlap_fused_rows", (k_overlap_fused_rows<false>), 8 * ((tiles + 7) / 8), PROBE_THREADS, v, qc, qs, qe, ids, n, vec, (long long)capacity, state, cols);
    HIP_TRY(hipMemcpyAsync(ctx->h_total, state, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipGetLastError());
    *n_pairs = ctx->h_total[0];
    if (ctx->h_total[1] != 0)
        return fail(IVJ_ECAPACITY, "output capacity " + std::to_string(capacity) + " < " + std::to_string(ctx->h_total[0]) + " rows");
    return IVJ_OK;
}

// counts32 != nullptr: int32 counts into counts32 (the per-probe exchange's wire column), `counts` unused
int count_overlaps_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int64_t* counts, int32_t* counts32 = nullptr) {
    // the kernel reads the joint grid only; the start table is needed by the (opt-in) bucketed form
    if (!ix->has_tables || opts->partition_mode == 1) IVJ_TRY(need_tables(ctx, ix));
    const int64_t n = probe->n;
    if (n == 0) return IVJ_OK;
    if (ix->n == 0) {
        if (counts32) HIP_TRY(hipMemsetAsync(counts32, 0, (size_t)n * 4, ctx->stream));
        else HIP_TRY(hipMemsetAsync(counts, 0, (size_t)n * 8, ctx->stream));
        return IVJ_OK;
    }
    IVJ_TRY(build_end_order(ctx, ix));
    // partition_mode 1 only: bucket the probes by genomic position, count in bucket order into scratch, bring the
    // counts back to probe order with the coalesced inverse permutation.  Not the default: with ONE record gather per
    // probe the bucketing + inverse permutation cost more than the L2 locality buys (100M x 5M: 4.1 ms plain, 5.2 ms
    // bucketed; nearest and coverage, with 3+ gathers per probe, do gain).
    const int32_t *qc = probe->contig, *qs = probe->start, *qe = probe->end;
    long long* o_counts = (long long*)counts;
    const bool bucketed = opts->partition_mode == 1 && ix->n > 0 && !counts32;
    if (bucketed) {
        ivj_side plain = *probe;
        plain.row_id = nullptr;
        IVJ_TRY(ensure_ov(ctx, n, 1));
        ctx->ov_n = -1;
        ivj_opts popts = *opts; popts.partition_mode = 1;
        IVJ_TRY(partition_probes(ctx, ix, &plain, &popts));
        qc = ctx->pt_c; qs = ctx->pt_s; qe = ctx->pt_e;
        IVJ_TRY(arena_reserve(ctx, align_up((size_t)n * 8) + 4096));
        o_counts = arena_take<long long>(ctx, n);
    }
    constexpr int NT = PROBE_THREADS * PROBE_ITEMS_LAT * COUNT_TILES_PER_WG;
    const int64_t tiles = (n + NT - 1) / NT;
    const bool vec = aligned16(qc) && aligned16(qs) && aligned16(qe);
    IndexView v = view_of(ix);
    const bool strict = opts->filter_op == IVJ_FILTER_STRICT;
    if (ix->n_contigs <= CM_LDS && !ctx->env_count_nolds) {
        if (strict) LAUNCH(ctx, "count_overlaps", (k_count_overlaps<true, PROBE_ITEMS_LAT, true>), tiles, PROBE_THREADS, v, qc, qs, qe, n, vec, o_counts, counts32, ctx->env_count_ablate);
        else LAUNCH(ctx, "count_overlaps", (k_count_overlaps<false, PROBE_ITEMS_LAT, true>), tiles, PROBE_THREADS, v, qc, qs, qe, n, vec, o_counts, counts32, ctx->env_count_ablate);
    } else {
        if (strict) LAUNCH(ctx, "count_overlaps", (k_count_overlaps<true, PROBE_ITEMS_LAT, false>), tiles, PROBE_THREADS, v, qc, qs, qe, n, vec, o_counts, counts32, ctx->env_count_ablate);
        else LAUNCH(ctx, "count_overlaps", (k_count_overlaps<false, PROBE_ITEMS_LAT, false>), tiles, PROBE_THREADS, v, qc, qs, qe, n, vec, o_counts, counts32, ctx->env_count_ablate);
    }
    HIP_TRY(hipGetLastError());
    if (bucketed) {
        UnpermuteCols uc{{o_counts, nullptr, nullptr}, {counts, nullptr, nullptr}, {8, 0, 0}, 1, nullptr};
        IVJ_TRY(unpermute(ctx, n, uc));
    }
    return IVJ_OK;
}

int nearest_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int32_t* idx, int64_t* dist, int32_t* nf) {
    IVJ_TRY(need_tables(ctx, ix));
    const int64_t n = probe->n;
    const int k = opts->nearest_k < 1 ? 1 : opts->nearest_k;
    if (n == 0) return IVJ_OK;
    const bool strict = opts->filter_op == IVJ_FILTER_STRICT;
    const bool k1 = k == 1 && opts->include_overlaps;
    if (k1) IVJ_TRY(build_argmax(ctx, ix));
    else { IVJ_TRY(build_end_order(ctx, ix)); IVJ_TRY(ensure_hier(ctx, ix)); }     // nearest_general lists overlapping rows with hier_walk_up
    // large probe sides: bucket them by genomic position first (every gather of the kernel then stays in the
    // XCD L2s); the kernels write each result to the probe's original row
    const int32_t *qc = probe->contig, *qs = probe->start, *qe = probe->end, *qrow = nullptr;
    // k = 1 over the nearest LINES: one 128-byte line fetch per probe, probes in input order (round 4; tools/micro/gather_probe.hip: a line
    // from beyond the L2 costs ~ 20 ps per probe whatever the table size, the bucketed two-gather form 36 ps with its partition and inverse
    // permutation).  The table costs 256 bytes per build row to write: auto takes it for an index beyond the L2s (>= 128 k rows) once it
    // exists or the probe side is >= 8 x the build side; partition_mode 1 / 2 keep their meaning (bucketed / plain two-gather kernel).
    if (k1 && ix->n > 0 && ix->n_contigs <= CM_LDS && ctx->env_nearest_lines != 0 &&
        (ctx->env_nearest_lines > 0 || ix->table_mode == 3 ||
         (opts->partition_mode == 0 && ix->table_mode == 0 && ix->n >= (128ll << 10) && (ix->has_lines || n >= 8 * ix->n))) &&
        (size_t)ix->bins_len * 64 <= ((size_t)16 << 30) && n <= 0x7fffffffll) {
        IVJ_TRY(build_lines(ctx, ix));
        IndexView v = view_of(ix);
        constexpr int NT = PROBE_THREADS * PROBE_ITEMS_LAT;
        const int64_t tiles = (n + NT - 1) / NT;
        const bool vec = aligned16(qc) && aligned16(qs) && aligned16(qe);
        // the probes a line cannot settle: one bit each in scratch, finished by the two-gather kernel over the same thread <-> probe mapping
        const int64_t nwf = tiles * (PROBE_THREADS / kWave);
        // (two masks per (wavefront, item): redo from the columns / settled but for the build row)
        const int64_t n_words = nwf * PROBE_ITEMS_LAT;
        IVJ_TRY(arena_reserve(ctx, align_up((size_t)n_words * 16) + 4096));
        unsigned long long* rest = arena_take<unsigned long long>(ctx, 2 * n_words);
        if (strict) LAUNCH(ctx, "nearest_k1_lines", (k_nearest_k1_lines<true, PROBE_ITEMS_LAT>), tiles, PROBE_THREADS, v, qc, qs, qe, n, vec, idx, (long long*)dist, nf, rest, n_words);
        else LAUNCH(ctx, "nearest_k1_lines", (k_nearest_k1_lines<false, PROBE_ITEMS_LAT>), tiles, PROBE_THREADS, v, qc, qs, qe, n, vec, idx, (long long*)dist, nf, rest, n_words);
        const int64_t rgrid = (n_words + REST_WORDS - 1) / REST_WORDS;
        if (strict) LAUNCH(ctx, "nearest_k1_rest", (k_nearest_k1_rest<true, PROBE_ITEMS_LAT>), rgrid, PROBE_THREADS, v, qc, qs, qe, n, n_words, (const unsigned long long*)rest, idx, (long long*)dist, nf);
        else LAUNCH(ctx, "nearest_k1_rest", (k_nearest_k1_rest<false, PROBE_ITEMS_LAT>), rgrid, PROBE_THREADS, v, qc, qs, qe, n, n_words, (const unsigned long long*)rest, idx, (long long*)dist, nf);
        HIP_TRY(hipGetLastError());
        return IVJ_OK;
    }
    if (want_partition(ix, n, opts) && ix->n > 0) {
        ivj_side plain = *probe;
        plain.row_id = nullptr;
        IVJ_TRY(ensure_ov(ctx, n, 1));
        ctx->ov_n = -1;
        ivj_opts popts = *opts; popts.partition_mode = 1;
        IVJ_TRY(partition_probes(ctx, ix, &plain, &popts));
        qc = ctx->pt_c; qs = ctx->pt_s; qe = ctx->pt_e; qrow = ctx->pt_row;
    }
    IndexView v = view_of(ix);
    if (k1) {
        constexpr int NT = PROBE_THREADS * PROBE_ITEMS_LAT;
        const int64_t tiles = (n + NT - 1) / NT;
        const bool vec = aligned16(qc) && aligned16(qs) && aligned16(qe);
        int32_t *o_idx = idx, *o_nf = nf;
        long long* o_dist = (long long*)dist;
        if (qrow) {
            // bucket-order results in scratch, then ONE coalesced inverse permutation of the three columns
            IVJ_TRY(arena_reserve(ctx, 2 * align_up((size_t)n * 4) + align_up((size_t)n * 8) + 4096));
            o_idx = arena_take<int32_t>(ctx, n); o_nf = arena_take<int32_t>(ctx, n); o_dist = arena_take<long long>(ctx, n);
        }
        if (strict) LAUNCH(ctx, "nearest_k1", (k_nearest_k1<true, PROBE_ITEMS_LAT>), 8 * ((tiles + 7) / 8), PROBE_THREADS, v, qc, qs, qe, n, vec, (const int32_t*)nullptr, o_idx, o_dist, o_nf, ctx->env_count_ablate);
        else LAUNCH(ctx, "nearest_k1", (k_nearest_k1<false, PROBE_ITEMS_LAT>), 8 * ((tiles + 7) / 8), PROBE_THREADS, v, qc, qs, qe, n, vec, (const int32_t*)nullptr, o_idx, o_dist, o_nf, ctx->env_count_ablate);
        if (qrow) {
            // n_found of k = 1 is "a row was found": derived from the row index while it is written
            UnpermuteCols uc{{o_idx, o_dist, nullptr}, {idx, dist, nullptr}, {4, 8, 0}, 2, nf};
            IVJ_TRY(unpermute(ctx, n, uc));
        }
    } else {
        if (strict) LAUNCH(ctx, "nearest_general", (k_nearest_general<true>), grid1d(n, PROBE_THREADS), PROBE_THREADS, v, qc, qs, qe, n, k, (int)opts->include_overlaps, qrow, idx, (long long*)dist, nf);
        else LAUNCH(ctx, "nearest_general", (k_nearest_general<false>), grid1d(n, PROBE_THREADS), PROBE_THREADS, v, qc, qs, qe, n, k, (int)opts->include_overlaps, qrow, idx, (long long*)dist, nf);
    }
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

// host side -> device copies of one side
struct DevSide {
    ivj_side s{nullptr, nullptr, nullptr, 0, nullptr};
    int32_t* buf = nullptr;
    ~DevSide() { if (buf) (void)hipFree(buf); }
};
int upload_side(ivj_ctx* ctx, const ivj_side* h, DevSide& d) {
    d.s.n = h->n;
    if (h->n == 0) return IVJ_OK;
    const size_t col = align_up((size_t)h->n * 4);
    hipError_t e = hipMalloc((void**)&d.buf, 3 * col);
    if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string("hipMalloc(side): ") + hipGetErrorString(e));
    int32_t* c = d.buf; int32_t* s = (int32_t*)((char*)d.buf + col); int32_t* en = (int32_t*)((char*)d.buf + 2 * col);
    // through the context's pinned staging slots (HostXfer): chunk i is copied into a slot while chunk i - 1 is on the wire
    const size_t nb = (size_t)h->n * 4;
    HostXfer copy(ctx->stream, &ctx->xfer);
    copy.h2d(c, h->contig, nb);
    copy.h2d(s, h->start, nb);
    copy.h2d(en, h->end, nb);
    HIP_TRY(copy.finish());
    d.s.contig = c; d.s.start = s; d.s.end = en;
    return IVJ_OK;
}

struct IndexHolder {
    ivj_index* ix = nullptr;
    ~IndexHolder() { if (ix) ivj_index_free(ix); }
};

}  // namespace
