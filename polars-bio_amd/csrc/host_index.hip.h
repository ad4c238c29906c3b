// host_index.hip.h -- build of the HBM index (radix sort, prefix max, direct-address tables) and its on-demand parts
// Part of the single translation unit ivjoin.hip (included there, in this order); not a stand-alone header.
#pragma once

namespace {

int need_tables(const ivj_index* ix) {
    if (ix->has_tables) return IVJ_OK;
    return fail(IVJ_ESTATE, "this index was built for merge / cluster only (with_end_order & 2): it has no lookup tables");
}

int build_end_order(ivj_ctx* ctx, ivj_index* ix) {
    if (ix->has_end_order) return IVJ_OK;
    const int64_t n = ix->n;
    if (n == 0) { ix->has_end_order = true; return IVJ_OK; }
    IVJ_TRY(arena_reserve(ctx, sort_scratch_bytes(n) + align_up((size_t)(scan_num_tiles(ix->bins_len) + 1) * 4) +
                               2 * align_up((size_t)ix->bins_len * 4) + 4096));
    SortBufs sb; take_sort_bufs(ctx, n, sb);
    uint32_t* bins_part = arena_take<uint32_t>(ctx, scan_num_tiles(ix->bins_len) + 1);
    uint32_t* jb_s = arena_take<uint32_t>(ctx, ix->bins_len);
    uint32_t* jb_e = arena_take<uint32_t>(ctx, ix->bins_len);
    LAUNCH(ctx, "end_keys", k_end_keys, grid1d(n, 256), 256, (const int2*)ix->ep, n, sb.kA, sb.vA);
    bool fl = radix_sort_pairs(ctx, sb, n, 32);
    if (fl) { std::swap(sb.kA, sb.kB); std::swap(sb.vA, sb.vB); }
    // contig of each sorted position, then the contig passes
    LAUNCH(ctx, "gather", k_gather_u32, grid1d(n, 256), 256, (const int32_t*)ix->b_contig, (const uint32_t*)sb.vA, n, sb.kA);
    fl = radix_sort_pairs(ctx, sb, n, bits_for((uint32_t)ix->n_contigs));
    const uint32_t* pos = fl ? sb.vB : sb.vA;
    const uint32_t* ckeys = fl ? sb.kB : sb.kA;
    LAUNCH(ctx, "end_finalize", k_end_finalize, grid1d(n, 256), 256, (const int2*)ix->ep, pos, n, ix->e_end, ix->e_pos);
    // direct-address table over the sorted ends (same segments as the start order)
    if (ix->n_contigs > 0) {
        LAUNCH(ctx, "contig_meta", k_contig_meta, grid1d(ix->n_contigs, 256), 256, (const int32_t*)ix->seg,
               (const int32_t*)ix->e_end, ix->n_contigs, ix->cmeta_e);
        HIP_TRY(hipMemsetAsync(ix->bins_e, 0, (size_t)ix->bins_len * 4, ctx->stream));
        LAUNCH(ctx, "bins_mark", k_bins_mark, grid1d(n, 256), 256, (const int32_t*)ix->e_end, (const int32_t*)ckeys, n,
               ix->n_contigs, (const int4*)ix->cmeta_e, ix->bins_e);
        device_scan<uint32_t, MaxOp, true>(ctx, "bins_scan", ix->bins_e, ix->bins_e, ix->bins_len, 0u, bins_part, (uint32_t*)nullptr);
        LAUNCH(ctx, "bins_records", k_bins_records, grid1d(ix->bins_len, 256), 256, (const uint32_t*)ix->bins_e, ix->bins_len,
               (const int32_t*)ix->e_end, (const int4*)ix->cmeta_e, ix->n_contigs, ix->brec_e);
        // joint grid for count_overlaps: the same bins for the start order and the end order
        // joint grid: two bins per build row, or ONE when that is what keeps the 32-byte records of a small build
        // side near an XCD's 4-MiB L2 (measured on 200 k rows: 3.11 -> 2.76 ms for 200 M probes)
        const int bins_per_row = ((size_t)n * 64 > (3u << 20) && (size_t)n * 32 <= (7u << 20)) ? 1 : 2;
        LAUNCH(ctx, "contig_meta", k_contig_meta_joint, grid1d(ix->n_contigs, 256), 256, (const int32_t*)ix->seg,
               (const int32_t*)ix->b_start, (const int32_t*)ix->e_end, ix->n_contigs, bins_per_row, ix->cmeta_j);
        HIP_TRY(hipMemsetAsync(jb_s, 0, (size_t)ix->bins_len * 4, ctx->stream));
        HIP_TRY(hipMemsetAsync(jb_e, 0, (size_t)ix->bins_len * 4, ctx->stream));
        LAUNCH(ctx, "bins_mark", k_bins_mark, grid1d(n, 256), 256, (const int32_t*)ix->b_start, (const int32_t*)ix->b_contig, n,
               ix->n_contigs, (const int4*)ix->cmeta_j, jb_s);
        LAUNCH(ctx, "bins_mark", k_bins_mark, grid1d(n, 256), 256, (const int32_t*)ix->e_end, (const int32_t*)ckeys, n,
               ix->n_contigs, (const int4*)ix->cmeta_j, jb_e);
        device_scan<uint32_t, MaxOp, true>(ctx, "bins_scan", jb_s, jb_s, ix->bins_len, 0u, bins_part, (uint32_t*)nullptr);
        device_scan<uint32_t, MaxOp, true>(ctx, "bins_scan", jb_e, jb_e, ix->bins_len, 0u, bins_part, (uint32_t*)nullptr);
        LAUNCH(ctx, "joint_records", k_joint_records, grid1d(ix->bins_len, 256), 256, (const uint32_t*)jb_s, (const uint32_t*)jb_e,
               ix->bins_len, (const int32_t*)ix->b_start, (const int32_t*)ix->e_end, (const int4*)ix->cmeta_j, ix->n_contigs, ix->crec);
    }
    ix->has_end_order = true;
    return IVJ_OK;
}

// pargmax[p] = position of the first row attaining the prefix max at p (nearest, k = 1)
int build_argmax(ivj_ctx* ctx, ivj_index* ix) {
    if (ix->has_argmax) return IVJ_OK;
    const int64_t n = ix->n;
    if (n == 0) { ix->has_argmax = true; return IVJ_OK; }
    IVJ_TRY(arena_reserve(ctx, align_up((size_t)(scan_num_tiles(n) + 1) * 4) + 4096));
    uint32_t* part = arena_take<uint32_t>(ctx, scan_num_tiles(n) + 1);
    LAUNCH(ctx, "pmax_change", k_pmax_change, grid1d(n, 256), 256, (const int2*)ix->ep, (const int32_t*)ix->b_contig, n, (uint32_t*)ix->pargmax);
    device_scan<uint32_t, MaxOp, true>(ctx, "argmax_scan", (uint32_t*)ix->pargmax, (uint32_t*)ix->pargmax, n, 0u, part, (uint32_t*)nullptr);
    LAUNCH(ctx, "nearest_records", k_nearest_records, grid1d(n + 1, 256), 256, (const int32_t*)ix->b_start, (const int2*)ix->ep,
           (const int32_t*)ix->b_row, (const int32_t*)ix->pargmax, n, ix->nrec);
    ix->has_argmax = true;
    return IVJ_OK;
}

// rec4[p] = {start, end, build row, prefix max}: built on demand for the join + materialisation path
int build_rec4(ivj_ctx* ctx, ivj_index* ix) {
    if (ix->has_rec4 || ix->n == 0) return IVJ_OK;
    LAUNCH(ctx, "rec4", k_rec4, grid1d(ix->n, 256), 256, (const int32_t*)ix->b_start, (const int2*)ix->ep, (const int32_t*)ix->b_row, ix->n, ix->rec4);
    HIP_TRY(hipGetLastError());
    ix->has_rec4 = true;
    return IVJ_OK;
}

// flat overlap path (flat.hip.h): per start bin the first position whose prefix max reaches it, interleaved with the
// bin table; rec4.  Filled on first use (dense results, partition_mode 5).
int build_flat(ivj_ctx* ctx, ivj_index* ix) {
    if (ix->has_flat || ix->n == 0 || ix->n_contigs <= 0) return IVJ_OK;
    IVJ_TRY(need_tables(ix));
    IVJ_TRY(arena_reserve(ctx, align_up((size_t)(scan_num_tiles(ix->bins_len) + 1) * 4) + 4096));
    uint32_t* part = arena_take<uint32_t>(ctx, scan_num_tiles(ix->bins_len) + 1);
    HIP_TRY(hipMemsetAsync(ix->lot, 0, (size_t)ix->bins_len * 4, ctx->stream));
    LAUNCH(ctx, "lot_mark", k_lot_mark, grid1d(ix->n, 256), 256, (const int2*)ix->ep, (const int32_t*)ix->b_contig, ix->n,
           ix->n_contigs, (const int4*)ix->cmeta, ix->lot);
    device_scan<uint32_t, MaxOp, true>(ctx, "lot_scan", ix->lot, ix->lot, ix->bins_len, 0u, part, (uint32_t*)nullptr);
    LAUNCH(ctx, "tab2", k_tab2, grid1d(ix->bins_len, 256), 256, (const uint32_t*)ix->bins, (const uint32_t*)ix->lot, ix->bins_len, ix->tab2);
    HIP_TRY(hipGetLastError());
    IVJ_TRY(build_rec4(ctx, ix));
    ix->has_flat = true;
    return IVJ_OK;
}

int index_build(ivj_ctx* ctx, const ivj_side* build, const ivj_opts* opts, int with_end_order, ivj_index** out) {
    // table offsets (2 a + 2 c) and slot counts are int32: 2 Nb + 2 n_contigs must stay below 2^31
    if (2 * build->n + 2 * (int64_t)opts->n_contigs + 64 > 0x7fffffffll)
        return fail(IVJ_EINVAL, "build side too large for the int32 direct-address table (2*rows + 2*contigs must be < 2^31)");
    ivj_index* ix = new ivj_index();
    ix->ctx = ctx; ix->device = ctx->device; ix->n = build->n; ix->n_contigs = opts->n_contigs; ix->table_mode = opts->table_mode;
    const int64_t n = build->n;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    auto cleanup = [&](int code) { ivj_index_free(ix); return code; };
    {
        const size_t col = align_up(nn * 4);
        const size_t nc = (size_t)opts->n_contigs;
        ix->bins_len = 2 * (int64_t)nn + 2 * (int64_t)nc + 16;
        const size_t small = align_up((nc + 2) * 4) + align_up(16) + 3 * align_up((nc + 1) * 32);   // seg, flags, cmeta, cmeta_e, cmeta_j
        const size_t flat_bytes = align_up((nn + 1) * 16) + 3 * align_up((size_t)ix->bins_len * 4);   // rec4, lot, tab2 (filled on demand)
        const size_t spl_bytes = align_up((size_t)SL_MAX_BUCKETS * 8) + align_up((size_t)SL_TAB_CONTIGS * 16) +
                                 align_up((size_t)(4 * SL_MAX_BUCKETS + SL_TAB_CONTIGS) * 4);
        const size_t need = spl_bytes + flat_bytes + 6 * col + align_up(nn * 8) + align_up((nn + 1) * 16) + 2 * align_up((size_t)ix->bins_len * 4) +
                            4 * align_up((size_t)ix->bins_len * 16) + small + 256;
        if (ctx->ix_cache && ctx->ix_cache_cap >= need) {
            ix->slab = ctx->ix_cache; ix->slab_cap = ctx->ix_cache_cap;
            ctx->ix_cache = nullptr; ctx->ix_cache_cap = 0;
        } else {
            hipError_t e = hipMalloc((void**)&ix->slab, need);
            if (e != hipSuccess) return cleanup(fail(IVJ_ENOMEM, std::string("hipMalloc(index): ") + hipGetErrorString(e)));
            ix->slab_cap = need;
        }
        char* p = ix->slab;
        ix->spl = (unsigned long long*)p;
        ix->sl_cm = (int4*)(p + align_up((size_t)SL_MAX_BUCKETS * 8));
        ix->sl_cell = (uint32_t*)((char*)ix->sl_cm + align_up((size_t)SL_TAB_CONTIGS * 16));
        p += spl_bytes;
        ix->ep = (int2*)p; p += align_up(nn * 8);
        ix->b_start = (int32_t*)p; p += col;
        ix->b_row = (int32_t*)p; p += col;
        ix->b_contig = (int32_t*)p; p += col;
        ix->e_end = (int32_t*)p; p += col;
        ix->e_pos = (int32_t*)p; p += col;
        ix->pargmax = (int32_t*)p; p += col;
        ix->nrec = (int4*)p; p += align_up((nn + 1) * 16);
        ix->bins = (uint32_t*)p; p += align_up((size_t)ix->bins_len * 4);
        ix->bins_e = (uint32_t*)p; p += align_up((size_t)ix->bins_len * 4);
        ix->brec = (int4*)p; p += align_up((size_t)ix->bins_len * 16);
        ix->brec_e = (int4*)p; p += align_up((size_t)ix->bins_len * 16);
        ix->crec = (int4*)p; p += align_up((size_t)ix->bins_len * 32);     // 32-byte joint records
        ix->rec4 = (int4*)p; p += align_up((nn + 1) * 16);
        {
            ix->lot = (uint32_t*)p; p += align_up((size_t)ix->bins_len * 4);
            ix->tab2 = (uint2*)p; p += 2 * align_up((size_t)ix->bins_len * 4);
        }
        char* small_base = p;
        ix->seg = (int32_t*)p; p += align_up((nc + 2) * 4);
        ix->flags = (int32_t*)p; p += align_up(16);
        ix->cmeta = (int4*)p; p += align_up((nc + 1) * 32);
        ix->cmeta_e = (int4*)p; p += align_up((nc + 1) * 32);
        ix->cmeta_j = (int4*)p;
        // seg, flags and cmeta start zeroed: an empty index answers every probe with "no rows"
        hipError_t e = hipMemsetAsync(small_base, 0, small, ctx->stream);
        if (e != hipSuccess) return cleanup(fail(IVJ_EHIP, std::string("hipMemsetAsync(index meta): ") + hipGetErrorString(e)));
    }
    if (n > 0) {
        const size_t comp_bytes = 2 * align_up((size_t)n * 8) + align_up((size_t)(scan_num_tiles(n) + 1) * 8) +
                                  align_up((size_t)(scan_num_tiles(ix->bins_len) + 1) * 4);
        int r = arena_reserve(ctx, sort_scratch_bytes(n) + comp_bytes + 4096);
        if (r != IVJ_OK) return cleanup(r);
        SortBufs sb; take_sort_bufs(ctx, n, sb);
        unsigned long long* comp = arena_take<unsigned long long>(ctx, n);
        unsigned long long* comp_max = arena_take<unsigned long long>(ctx, n);
        unsigned long long* comp_part = arena_take<unsigned long long>(ctx, scan_num_tiles(n) + 1);
        uint32_t* bins_part = arena_take<uint32_t>(ctx, scan_num_tiles(ix->bins_len) + 1);
        // 1. stable sort by start (row ids as payload), 2. stable sort by contig id
        LAUNCH(ctx, "sort_keys", k_iota_flip, grid1d(n, 256), 256, build->start, n, sb.kA, sb.vA);
        bool fl = radix_sort_pairs(ctx, sb, n, 32);
        if (fl) { std::swap(sb.kA, sb.kB); std::swap(sb.vA, sb.vB); }
        LAUNCH(ctx, "gather", k_gather_contig, grid1d(n, 256), 256, build->contig, (const uint32_t*)sb.vA, n, opts->n_contigs, sb.kA);
        fl = radix_sort_pairs(ctx, sb, n, bits_for((uint32_t)opts->n_contigs));
        const uint32_t* ckeys = fl ? sb.kB : sb.kA;
        const uint32_t* rows = fl ? sb.vB : sb.vA;
        // 3. sorted columns, segment offsets, (contig,end) composites; 4. prefix max; 5. interleave (end, pmax)
        LAUNCH(ctx, "index_finalize", k_index_finalize, grid1d(n, 256), 256, build->start, build->end, rows, ckeys, build->row_id, n,
               opts->n_contigs, ix->b_start, ix->b_row, ix->b_contig, comp, ix->seg, ix->flags);
        device_scan<unsigned long long, MaxOp, true>(ctx, "pmax_scan", comp, comp_max, n, 0ull, comp_part,
                                                      (unsigned long long*)nullptr);
        LAUNCH(ctx, "emit_ep", k_emit_ep, grid1d(n, 256), 256, (const unsigned long long*)comp,
               (const unsigned long long*)comp_max, n, ix->ep);
        // 6. direct-address table over start
        ix->has_tables = !(with_end_order & 2);
        if (opts->n_contigs > 0 && ix->has_tables) {
            LAUNCH(ctx, "contig_meta", k_contig_meta, grid1d(opts->n_contigs, 256), 256, (const int32_t*)ix->seg,
                   (const int32_t*)ix->b_start, opts->n_contigs, ix->cmeta);
            hipError_t me = hipMemsetAsync(ix->bins, 0, (size_t)ix->bins_len * 4, ctx->stream);
            if (me != hipSuccess) return cleanup(fail(IVJ_EHIP, std::string("hipMemsetAsync(bins): ") + hipGetErrorString(me)));
            LAUNCH(ctx, "bins_mark", k_bins_mark, grid1d(n, 256), 256, (const int32_t*)ix->b_start, (const int32_t*)ix->b_contig, n,
                   opts->n_contigs, (const int4*)ix->cmeta, ix->bins);
            device_scan<uint32_t, MaxOp, true>(ctx, "bins_scan", ix->bins, ix->bins, ix->bins_len, 0u, bins_part, (uint32_t*)nullptr);
            LAUNCH(ctx, "bins_records", k_bins_records, grid1d(ix->bins_len, 256), 256, (const uint32_t*)ix->bins, ix->bins_len,
                   (const int32_t*)ix->b_start, (const int4*)ix->cmeta, opts->n_contigs, ix->brec);
        }
        // 7. the flat overlap path's arrays (lot / tab2 / rec4) are filled on first use: build_flat
        if (opts->partition_mode == 5) { r = build_flat(ctx, ix); if (r != IVJ_OK) return cleanup(r); }
        if (with_end_order & 1) { r = build_end_order(ctx, ix); if (r != IVJ_OK) return cleanup(r); }
    } else {
        ix->has_end_order = true;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cleanup(fail(IVJ_EHIP, std::string("index build launch: ") + hipGetErrorString(e)));
    ctx->live.push_back(ix);
    *out = ix;
    return IVJ_OK;
}

}  // namespace
