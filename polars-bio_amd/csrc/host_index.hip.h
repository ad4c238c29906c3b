// host_index.hip.h -- build of the HBM index (radix sort, prefix max, direct-address tables) and its on-demand parts
// Part of the single translation unit ivjoin.hip (included there, in this order); not a stand-alone header.
#pragma once

namespace {

int build_tables(ivj_ctx* ctx, ivj_index* ix);
// contig-aligned slice path (host_cslice.hip.h): geometry and per-index arrays
bool cs_geom(int64_t n, int nc, int want_rows, CsGeom& g);
size_t cs_index_bytes(const CsGeom& g);
void cs_index_carve(ivj_index* ix, char* p);

// The direct-address tables (bins / brec over the starts) are built on first use: the slice path of pb.overlap never needs
// them, the window-scan kernels, nearest, count_overlaps, coverage and subtract do.
// the sorted ends and their block maxima, level by level (index_view.hip.h: windows that run on); filled on first use
int ensure_hier(ivj_ctx* ctx, ivj_index* ix) {
    if (ix->hier_built || ix->n <= 0) return IVJ_OK;
    const HierShape h = hier_shape(ix->n);
    // blocks past the rows (the pads of levels 0 .. 2) are written by the same launch: one workgroup per 4096 padded rows
    const int64_t wgs = (((ix->n + 15) & ~(int64_t)15) + HIER_WG_ROWS - 1) / HIER_WG_ROWS;
    LAUNCH(ctx, "hier_low", k_hier_low, (unsigned)wgs, 256, (const int2*)ix->ep, ix->n, ix->hier, h);
    if (h.nlev >= 3) LAUNCH(ctx, "hier_high", k_hier_high, 1, 256, ix->hier, h, wgs);
    HIP_TRY(hipGetLastError());
    ix->hier_built = true;
    return IVJ_OK;
}

int need_tables(ivj_ctx* ctx, ivj_index* ix) {
    if (!ix->has_tables) return fail(IVJ_ESTATE, "this index was built for merge / cluster only (with_end_order & 2): it has no lookup tables");
    if (ix->tables_built) return IVJ_OK;
    return build_tables(ctx, ix);
}

// ---- the 11-bit LSD sort of 16-byte records {key, payload, row, contig} by (contig, key - min key) (onesweep.hip.h) -------------
// Geometry + scratch of one sort; the scratch lives in the context's arena (the caller reserves os_sort_bytes and more).
struct OsSort {
    int64_t n = 0, chunk = 0, hist_len = 0, hs_tiles = 0, tiles = 0;
    int nc = 0, cbits = 0, passes_max = 0, nchunks = 0;
    size_t z_meta = 0, z_fin = 0, z_hs = 0, z_hist = 0, z_tick = 0, zero_bytes = 0;
    char* z = nullptr;
    int4 *recA = nullptr, *recB = nullptr;
    OsMeta* meta = nullptr;
    unsigned long long* st_fin = nullptr;
};
size_t os_sort_plan(OsSort& S, int64_t n, int nc) {
    S.n = n; S.nc = nc;
    S.cbits = os_bits_for((uint32_t)nc);                              // contig ids 0 .. nc (nc = rows outside the dictionary)
    S.passes_max = (32 + S.cbits + OS_BITS - 1) / OS_BITS;
    S.tiles = (n + OS_TILE - 1) / OS_TILE;
    // chunks of the passes: ONE workgroup per CU (the pass kernels take the whole LDS), whole sub-tiles: 5 M rows = 245 workgroups
    // of 5 sub-tiles in one round instead of 407 of 3 in two (the second round only 59 % full)
    S.chunk = ((n + 255) / 256 + OS_TILE - 1) / OS_TILE * OS_TILE;
    if (S.chunk < OS_TILE) S.chunk = OS_TILE;
    S.nchunks = (int)((n + S.chunk - 1) / S.chunk);
    S.hist_len = (int64_t)OS_RADIX * S.nchunks;
    S.hs_tiles = (S.hist_len + LB_TILE - 1) / LB_TILE;
    S.z_meta = align_up(sizeof(OsMeta)); S.z_fin = align_up((size_t)S.tiles * 8);
    S.z_hs = align_up((size_t)S.hs_tiles * 8) * (size_t)S.passes_max; S.z_hist = align_up((size_t)S.hist_len * 4) * (size_t)S.passes_max;
    S.z_tick = align_up((size_t)(S.passes_max + 2) * 4);
    S.zero_bytes = S.z_meta + S.z_fin + S.z_hs + S.z_hist + S.z_tick;
    return S.zero_bytes + 2 * align_up((size_t)n * 16);
}
void os_sort_take(ivj_ctx* ctx, OsSort& S) {
    S.z = arena_take<char>(ctx, S.zero_bytes);
    S.recA = arena_take<int4>(ctx, S.n);
    S.recB = arena_take<int4>(ctx, S.n);
    S.meta = (OsMeta*)S.z;
    S.st_fin = (unsigned long long*)(S.z + S.z_meta);
}
// min / max -> per pass [digit histogram per chunk, look-back scan, LDS-staged stable scatter of the records].  The sorted
// records end up in recA or recB depending on the number of passes the key width needs (decided on the device: the consumer
// kernel picks the buffer from meta, like k_ix_final).  `payload` travels in the record's .y, the row id (or i) in .z.
int os_sort_run(ivj_ctx* ctx, OsSort& S, const int32_t* contig, const int32_t* key, const int32_t* payload, const int32_t* row_id) {
    const int64_t n = S.n;
    char* st_hs = S.z + S.z_meta + S.z_fin;
    char* hists = st_hs + S.z_hs;
    uint32_t* tickets = (uint32_t*)(hists + S.z_hist);                // look-back scan tickets: one per pass
    HIP_TRY(hipMemsetAsync(S.z, 0, S.zero_bytes, ctx->stream));
    const unsigned sgrid = (unsigned)(S.tiles < 512 ? S.tiles : 512);
    LAUNCH(ctx, "ix_minmax", k_ix_minmax, sgrid, OS_THREADS, key, payload, contig, n, S.nc, S.meta);
    auto hist_of = [&](int p) { return (uint32_t*)(hists + (size_t)p * align_up((size_t)S.hist_len * 4)); };
    const size_t pass_lds = (size_t)os_pass_lds().total;
    if (!ctx->os_attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_os_scatter<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pass_lds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_os_scatter<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pass_lds));
        ctx->os_attr_set = true;
    }
    for (int p = 0; p < S.passes_max; ++p) {
        // a pass whose digit lies beyond the key bits exits at once (device-side decision: the key width depends on the data);
        // its scan then runs over a zero histogram
        const int4* src = (p & 1) ? S.recA : S.recB;                 // pass p writes buffer p & 1 (A, B, A, ...), reads the other
        int4* dst = (p & 1) ? S.recB : S.recA;
        if (p == 0) LAUNCH(ctx, "ix_hist", (k_os_hist<true>), S.nchunks, OS_THREADS, contig, key, src, n, S.nc, S.cbits, p, (const OsMeta*)S.meta, (int)S.chunk, S.nchunks, hist_of(p));
        else LAUNCH(ctx, "ix_hist", (k_os_hist<false>), S.nchunks, OS_THREADS, contig, key, src, n, S.nc, S.cbits, p, (const OsMeta*)S.meta, (int)S.chunk, S.nchunks, hist_of(p));
        LAUNCH(ctx, "ix_scan", (k_scan_lb_u32<SumOp, true>), S.hs_tiles, OS_THREADS, hist_of(p), S.hist_len, 0u, tickets + p,
               (unsigned long long*)(st_hs + (size_t)p * align_up((size_t)S.hs_tiles * 8)));
        t_begin(ctx, "ix_pass");
        if (p == 0)
            hipLaunchKernelGGL((k_os_scatter<true>), dim3((unsigned)S.nchunks), dim3(OS_THREADS), pass_lds, ctx->stream, contig, key, payload,
                               row_id, src, dst, n, S.nc, S.cbits, p, (const OsMeta*)S.meta, (int)S.chunk, S.nchunks, (const uint32_t*)hist_of(p));
        else
            hipLaunchKernelGGL((k_os_scatter<false>), dim3((unsigned)S.nchunks), dim3(OS_THREADS), pass_lds, ctx->stream, contig, key, payload,
                               row_id, src, dst, n, S.nc, S.cbits, p, (const OsMeta*)S.meta, (int)S.chunk, S.nchunks, (const uint32_t*)hist_of(p));
        t_end(ctx);
    }
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

int build_end_order(ivj_ctx* ctx, ivj_index* ix) {
    if (ix->has_end_order) return IVJ_OK;
    const int64_t n = ix->n;
    if (n == 0) { ix->has_end_order = true; return IVJ_OK; }
    // ends sorted by (contig, end, position): the same three 11-bit passes as the start order (keys relative to the smallest
    // end; the record's row field carries the position), then ONE kernel unpacks e_end / e_pos / the contig per position.
    // (Round 1: 4 + 1 eight-bit passes with 3-launch scans = 25 launches, 0.3 ms even for 200 k rows.)
    OsSort S;
    const int64_t lb_tiles = (ix->bins_len + LB_TILE - 1) / LB_TILE;
    const size_t zb = align_up((size_t)lb_tiles * 8) + align_up(16);          // look-back scan status + ticket, per scan
    IVJ_TRY(arena_reserve(ctx, os_sort_plan(S, n, ix->n_contigs) + align_up((size_t)n * 4) + 2 * zb +
                               2 * align_up((size_t)ix->bins_len * 4) + 4096));
    os_sort_take(ctx, S);
    int32_t* ekey = arena_take<int32_t>(ctx, n);
    char* zscan = arena_take<char>(ctx, 2 * zb);
    uint32_t* jb_s = arena_take<uint32_t>(ctx, ix->bins_len);
    uint32_t* jb_e = arena_take<uint32_t>(ctx, ix->bins_len);
    HIP_TRY(hipMemsetAsync(zscan, 0, 2 * zb, ctx->stream));
    auto lb_max_scan = [&](uint32_t* data, int k) {
        char* z = zscan + (size_t)k * zb;
        LAUNCH(ctx, "bins_scan", (k_scan_lb_u32<MaxOp, false>), lb_tiles, OS_THREADS, data, ix->bins_len, 0u,
               (uint32_t*)(z + align_up((size_t)lb_tiles * 8)), (unsigned long long*)z);
    };
    LAUNCH(ctx, "end_keys", k_end_column, grid1d(n, 256), 256, (const int2*)ix->ep, n, ekey);
    IVJ_TRY(os_sort_run(ctx, S, (const int32_t*)ix->b_contig, ekey, ekey, nullptr));
    LAUNCH(ctx, "end_finalize", k_end_unpack, grid1d(n, 256), 256, (const int4*)S.recA, (const int4*)S.recB, n, S.cbits, (const OsMeta*)S.meta,
           ix->e_end, ix->e_pos);
    // (both orders are sorted by contig first and share the segment offsets: the contig of end-sorted position p is b_contig[p])
    const int32_t* ckeys = ix->b_contig;
    if (ix->n_contigs > 0) {
        // joint grid for count_overlaps: the same bins for the start order and the end order
        // joint grid of 16-byte records: two bins per build row (200 k rows, 200 M probes: 1.96 ms against 2.07 ms with one
        // bin per row, although only the latter table fits an XCD's L2: fewer third-row searches matter more)
        int bins_per_row = 2;
        if (ctx->env_joint_bins == 1 || ctx->env_joint_bins == 2) bins_per_row = ctx->env_joint_bins;
        LAUNCH(ctx, "contig_meta", k_contig_meta_joint, grid1d(ix->n_contigs, 256), 256, (const int32_t*)ix->seg,
               (const int32_t*)ix->b_start, (const int32_t*)ix->e_end, ix->n_contigs, bins_per_row, ix->cmeta_j);
        HIP_TRY(hipMemsetAsync(jb_s, 0, (size_t)ix->bins_len * 4, ctx->stream));
        HIP_TRY(hipMemsetAsync(jb_e, 0, (size_t)ix->bins_len * 4, ctx->stream));
        LAUNCH(ctx, "bins_mark", k_bins_mark, grid1d(n, 256), 256, (const int32_t*)ix->b_start, (const int32_t*)ix->b_contig, n,
               ix->n_contigs, (const int4*)ix->cmeta_j, jb_s);
        LAUNCH(ctx, "bins_mark", k_bins_mark, grid1d(n, 256), 256, (const int32_t*)ix->e_end, (const int32_t*)ckeys, n,
               ix->n_contigs, (const int4*)ix->cmeta_j, jb_e);
        lb_max_scan(jb_s, 0);
        lb_max_scan(jb_e, 1);
        LAUNCH(ctx, "joint_records", k_joint_records, grid1d(ix->bins_len, 256), 256, (const uint32_t*)jb_s, (const uint32_t*)jb_e,
               ix->bins_len, (const int32_t*)ix->b_start, (const int32_t*)ix->e_end, (const int4*)ix->cmeta_j, ix->n_contigs, ix->crec);
    }
    ix->has_end_order = true;
    return IVJ_OK;
}

// direct-address table over the sorted ends (same segments as the start order): only the flat overlap kernel's rank formula
// reads it, so it is built on that kernel's first use
int build_end_table(ivj_ctx* ctx, ivj_index* ix) {
    IVJ_TRY(build_end_order(ctx, ix));
    if (ix->has_end_table || ix->n == 0 || ix->n_contigs <= 0) { ix->has_end_table = true; return IVJ_OK; }
    const int64_t n = ix->n;
    const int64_t lb_tiles = (ix->bins_len + LB_TILE - 1) / LB_TILE;
    const size_t zb = align_up((size_t)lb_tiles * 8) + align_up(16);
    IVJ_TRY(arena_reserve(ctx, zb + 4096));
    char* z = arena_take<char>(ctx, zb);
    HIP_TRY(hipMemsetAsync(z, 0, zb, ctx->stream));
    LAUNCH(ctx, "contig_meta", k_contig_meta, grid1d(ix->n_contigs, 256), 256, (const int32_t*)ix->seg,
           (const int32_t*)ix->e_end, ix->n_contigs, ix->cmeta_e);
    HIP_TRY(hipMemsetAsync(ix->bins_e, 0, (size_t)ix->bins_len * 4, ctx->stream));
    LAUNCH(ctx, "bins_mark", k_bins_mark, grid1d(n, 256), 256, (const int32_t*)ix->e_end, (const int32_t*)ix->b_contig, n,
           ix->n_contigs, (const int4*)ix->cmeta_e, ix->bins_e);
    LAUNCH(ctx, "bins_scan", (k_scan_lb_u32<MaxOp, false>), lb_tiles, OS_THREADS, ix->bins_e, ix->bins_len, 0u,
           (uint32_t*)(z + align_up((size_t)lb_tiles * 8)), (unsigned long long*)z);
    LAUNCH(ctx, "bins_records", k_bins_records, grid1d(ix->bins_len, 256), 256, (const uint32_t*)ix->bins_e, ix->bins_len,
           (const int32_t*)ix->e_end, (const int4*)ix->cmeta_e, ix->n_contigs, ix->brec_e);
    HIP_TRY(hipGetLastError());
    ix->has_end_table = true;
    return IVJ_OK;
}

// pargmax[p] = position of the first row attaining the prefix max at p (nearest, k = 1)
int build_argmax(ivj_ctx* ctx, ivj_index* ix) {
    if (ix->has_argmax) return IVJ_OK;
    const int64_t n = ix->n;
    if (n == 0) { ix->has_argmax = true; return IVJ_OK; }
    LAUNCH(ctx, "pmax_change", k_pmax_change, grid1d(n, 256), 256, (const int2*)ix->ep, (const int32_t*)ix->b_contig, n, (uint32_t*)ix->pargmax);
    IVJ_TRY((lb_scan_u32<MaxOp, false>(ctx, "argmax_scan", (uint32_t*)ix->pargmax, n, 0u)));
    LAUNCH(ctx, "nearest_records", k_nearest_records, grid1d(n + 1, 256), 256, (const int32_t*)ix->b_start, (const int2*)ix->ep,
           (const int32_t*)ix->b_row, (const int32_t*)ix->b_contig, (const int32_t*)ix->pargmax, n, ix->nrec);
    ix->has_argmax = true;
    return IVJ_OK;
}

// nearest lines (k_nearest_lines): 64 bytes per table slot (128 per build row) in an allocation of its own (kept by the context between indexes like
// the index slab), on first use by nearest_dev
int build_lines(ivj_ctx* ctx, ivj_index* ix) {
    if (ix->has_lines) return IVJ_OK;
    if (!ix->has_tables) return fail(IVJ_ESTATE, "this index was built for merge / cluster only (with_end_order & 2): it has no lookup tables");
    const size_t need = (size_t)ix->bins_len * 64;
    if (ix->nline_cap < need) {
        if (ix->nline) { (void)hipFree(ix->nline); ix->nline = nullptr; ix->nline_cap = 0; }
        if (ctx->nl_cache && ctx->nl_cache_cap >= need) {
            ix->nline = reinterpret_cast<int4*>(ctx->nl_cache); ix->nline_cap = ctx->nl_cache_cap;
            ctx->nl_cache = nullptr; ctx->nl_cache_cap = 0;
        } else {
            hipError_t e = hipMalloc((void**)&ix->nline, need);
            if (e != hipSuccess) { ix->nline = nullptr; return fail(IVJ_ENOMEM, std::string("hipMalloc(nearest lines): ") + hipGetErrorString(e)); }
            ix->nline_cap = need;
        }
    }
    // (Round 6, measured and dropped: the two table chains -- bins on the context's stream, pmax_change -> argmax_scan -> nearest_records
    // + bins_records on a second stream -- ran config 4 at 1.598 ms against 1.600 on one stream: these 10 - 30 us kernels are already
    // throughput-bound on their 16 - 64 MB of traffic, not launch- or latency-bound; profiles/r06/ab_nearest_side_stream_*.json)
    IVJ_TRY(need_tables(ctx, ix));
    IVJ_TRY(build_argmax(ctx, ix));
    LAUNCH(ctx, "nearest_lines", k_nearest_lines, grid1d(ix->bins_len * 4, 256), 256, (const uint32_t*)ix->bins, (const int4*)ix->cmeta, ix->n_contigs, (const int4*)ix->nrec,
           (const int32_t*)ix->b_start, (const int2*)ix->ep, (const int32_t*)ix->b_row, ix->bins_len, ix->n, ix->nline);
    HIP_TRY(hipGetLastError());
    ix->has_lines = true;
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
    IVJ_TRY(need_tables(ctx, ix));
    HIP_TRY(hipMemsetAsync(ix->lot, 0, (size_t)ix->bins_len * 4, ctx->stream));
    LAUNCH(ctx, "lot_mark", k_lot_mark, grid1d(ix->n, 256), 256, (const int2*)ix->ep, (const int32_t*)ix->b_contig, ix->n,
           ix->n_contigs, (const int4*)ix->cmeta, ix->lot);
    IVJ_TRY((lb_scan_u32<MaxOp, false>(ctx, "lot_scan", ix->lot, ix->bins_len, 0u)));
    LAUNCH(ctx, "tab2", k_tab2, grid1d(ix->bins_len, 256), 256, (const uint32_t*)ix->bins, (const uint32_t*)ix->lot, ix->bins_len, ix->tab2);
    HIP_TRY(hipGetLastError());
    IVJ_TRY(build_rec4(ctx, ix));
    ix->has_flat = true;
    return IVJ_OK;
}

// round-2 build: sort of {start, end, row, contig} -> index arrays (look-back prefix max), segment offsets
int index_sort_v2(ivj_ctx* ctx, ivj_index* ix, const ivj_side* build, const ivj_opts* opts) {
    OsSort S;
    IVJ_TRY(arena_reserve(ctx, os_sort_plan(S, build->n, opts->n_contigs) + 4096));
    os_sort_take(ctx, S);
    IVJ_TRY(os_sort_run(ctx, S, build->contig, build->start, build->end, build->row_id));
    LAUNCH(ctx, "ix_final", k_ix_final, S.tiles, OS_THREADS, (const int4*)S.recA, (const int4*)S.recB, S.n, S.nc, S.cbits, S.meta, S.st_fin, ix->b_start, ix->ep, ix->b_row,
           ix->b_contig, ix->seg, ix->flags);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

// round-5 build (ixsort3.hip.h): per-contig extremes -> ONE balanced bucket pass over HBM -> LDS sort per bucket that writes the
// index arrays.  *done = false hands the build to index_sort_v2 -- more contig keys than the LDS tables hold, linear keys beyond
// 32 bits, or a bucket above V3_CAP rows (clustered build sides): known from 8 bytes read back while the bucket pass runs, so exactness
// never rests on the balance.  Auto: 128 k .. 7 M rows (below, the launches are the cost either way; above, 2048 buckets of V3_CAP
// rows cannot hold the rows); IVJ_IX_V3 = 0 / 1 forces the choice (A/B runs, tests).
bool ix3_wanted(const ivj_ctx* ctx, int64_t n, int nc) {
    if (nc + 1 > V3_MAX_KEYS || n <= 0 || n > (int64_t)V3_CAP * V3_BUCKETS) return false;
    if (ctx->env_ix_v3 >= 0) return ctx->env_ix_v3 != 0;
    // a context whose last balanced builds were handed back (clustered build sides: every hand-over throws a scatter pass away) skips the
    // balanced build for a while: after 2 hand-overs in a row only every 16th build tries it again
    if (ctx->ix3_fallback_streak >= 2 && (ctx->ix3_builds_since_fallback & 15) != 15) return false;
    return n >= (128ll << 10) && n <= (7ll << 20);
}
struct V3Plan { int64_t chunk, hist_len, hs_tiles; int nchunks; size_t z_meta, z_st, z_hs, z_tick, zero_bytes; };
V3Plan v3_plan(int64_t n) {
    V3Plan p;
    p.chunk = ((n + 255) / 256 + OS_TILE - 1) / OS_TILE * OS_TILE;             // one workgroup per CU, whole sub-tiles (as os_sort_plan)
    if (p.chunk < OS_TILE) p.chunk = OS_TILE;
    p.nchunks = (int)((n + p.chunk - 1) / p.chunk);
    p.hist_len = (int64_t)V3_BUCKETS * p.nchunks;
    p.hs_tiles = (p.hist_len + LB_TILE - 1) / LB_TILE;
    p.z_meta = align_up(sizeof(V3Meta)); p.z_st = align_up((size_t)V3_BUCKETS * 8); p.z_hs = align_up((size_t)p.hs_tiles * 8); p.z_tick = align_up(16);
    p.zero_bytes = p.z_meta + p.z_st + p.z_hs + p.z_tick;
    return p;
}
int index_sort_v3(ivj_ctx* ctx, ivj_index* ix, const ivj_side* build, const ivj_opts* opts, bool* done) {
    *done = false;
    const int64_t n = build->n;
    const int nc = opts->n_contigs;
    const int64_t tiles = (n + OS_TILE - 1) / OS_TILE;
    const V3Plan VP = v3_plan(n);
    const int64_t chunk = VP.chunk, hist_len = VP.hist_len, hs_tiles = VP.hs_tiles;
    const int nchunks = VP.nchunks;
    const size_t z_meta = VP.z_meta, z_st = VP.z_st, z_hs = VP.z_hs, zero_bytes = VP.zero_bytes;
    OsSort S;                                                                   // (the arena must also hold the fallback's scratch: reserve the larger of the two once)
    const size_t v2_bytes = os_sort_plan(S, n, nc);
    // the words that start zeroed (V3Meta, status words of the two look-back chains, tickets) live in the index slab's zeroed head
    // when the slab was carved for this build (ix->ix3_z: one fill per index build instead of two), else in the arena
    const bool own_z = ix->ix3_z == nullptr;
    IVJ_TRY(arena_reserve(ctx, std::max(v2_bytes, (own_z ? zero_bytes : 0) + align_up((size_t)hist_len * 4) + align_up((size_t)n * 16)) + 4096));
    char* z = own_z ? arena_take<char>(ctx, zero_bytes) : ix->ix3_z;
    uint32_t* hist = arena_take<uint32_t>(ctx, (size_t)hist_len);
    int4* recs = arena_take<int4>(ctx, (size_t)n);
    V3Meta* meta = (V3Meta*)z;
    unsigned long long* st_local = (unsigned long long*)(z + z_meta);
    unsigned long long* st_scan = (unsigned long long*)(z + z_meta + z_st);
    uint32_t* tick_scan = (uint32_t*)(z + z_meta + z_st + z_hs);
    if (own_z) HIP_TRY(hipMemsetAsync(z, 0, zero_bytes, ctx->stream));
    const size_t pass_lds = (size_t)v3_pass_lds().total;
    if (!ctx->ix3_attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_v3_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pass_lds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_v3_local<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_v3_local<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        ctx->ix3_attr_set = true;
    }
    const unsigned sgrid = (unsigned)(tiles < 512 ? tiles : 512);
    LAUNCH(ctx, "ix3_stats", k_v3_stats, sgrid, OS_THREADS, build->start, build->end, build->contig, n, nc, meta);
    LAUNCH(ctx, "ix3_hist", k_v3_hist, nchunks, OS_THREADS, build->contig, build->start, n, nc, meta, (int)chunk, nchunks, hist);
    LAUNCH(ctx, "ix_scan", (k_scan_lb_u32<SumOp, true>), hs_tiles, OS_THREADS, hist, hist_len, 0u, tick_scan, st_scan);
    // (the check also picks the merge shift of the local kernel: 2^ms adjacent buckets per workgroup where they fit; IVJ_IX_MERGE pins an upper bound)
    const int ms_max = ctx->env_ix_merge >= 0 ? std::min(ctx->env_ix_merge, V3_MAX_MERGE) : V3_MAX_MERGE;
    const uint32_t hw_seq = ctx->hw ? ++ctx->hw_seq : 0u;
    LAUNCH(ctx, "ix3_check", k_v3_check, 1, OS_THREADS, (const uint32_t*)hist, nchunks, n, ms_max, meta, ctx->hw_dev, hw_seq);
    // {bad, max_bucket} travel to the host WHILE the bucket pass runs: the check kernel stores them into the host words (or, without
    // those, a copy of the two adjacent V3Meta words is queued in front of the pass); the host waits for the event behind the check
    // only -- by the time it knows, the pass is still running and the local kernel is queued behind it without a bubble.  (A build
    // that falls back has run the pass for nothing: rare, and exactness does not depend on it.)
    if (!ctx->hw) HIP_TRY(hipMemcpyAsync(ctx->h_total + 6, &meta->bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (!ctx->ix3_event) HIP_TRY(hipEventCreateWithFlags(&ctx->ix3_event, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ctx->ix3_event, ctx->stream));
    t_begin(ctx, "ix3_pass");
    hipLaunchKernelGGL(k_v3_scatter, dim3((unsigned)nchunks), dim3(OS_THREADS), pass_lds, ctx->stream, build->contig, build->start, build->end, build->row_id,
                       recs, n, nc, meta, (int)chunk, nchunks, (const uint32_t*)hist);
    t_end(ctx);
    HIP_TRY(hipGetLastError());
    HIP_TRY(wait_event(ctx, ctx->ix3_event));
    uint32_t* hv = reinterpret_cast<uint32_t*>(ctx->h_total + 6);
    if (ctx->hw) {
        const volatile uint32_t* w = reinterpret_cast<volatile uint32_t*>(ctx->hw);
        if (w[2] == hw_seq) { hv[0] = w[0]; hv[1] = w[1]; }
        else {                                                                 // the words did not arrive: by copy (the pass has to finish first)
            ++ctx->hw_misses;
            HIP_TRY(hipMemcpyAsync(ctx->h_total + 6, &meta->bad, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
    }
    const uint32_t max_bucket = hv[1] & 0xffffffu;
    const int ms = (int)(hv[1] >> 24);
    if (hv[0] != 0u || max_bucket > (uint32_t)V3_CAP) { ++ctx->ix3_fallbacks; ++ctx->ix3_fallback_streak; ctx->ix3_builds_since_fallback = 0; return IVJ_OK; }
    ctx->ix3_fallback_streak = 0;
    // the local kernel's LDS follows the LARGEST bucket (known now): rows staged in LDS while two workgroups still share a CU
    // (<= 2048 rows), beyond that the rows are read again from their L2-resident records (IVJ_IX_STAGE = 0 / 1 forces either form)
    const int cap = (int)std::max<uint32_t>(1024u, (max_bucket + 1023u) / 1024u * 1024u);
    const int bin_bits = cap <= 1024 ? 10 : (cap <= 2048 ? 11 : V3_MAX_BIN_BITS);
    const bool stage = ctx->env_ix_stage >= 0 ? ctx->env_ix_stage != 0 : cap <= 2048;
    const size_t local_lds = (size_t)v3_local_lds(cap, 1 << bin_bits, stage).total;
    t_begin(ctx, "ix3_local");
    if (stage) hipLaunchKernelGGL(k_v3_local<true>, dim3(V3_BUCKETS >> ms), dim3(OS_THREADS), local_lds, ctx->stream, (const int4*)recs, (const uint32_t*)hist, nchunks, n, nc, cap, bin_bits, ms,
                                  meta, st_local, ix->b_start, ix->ep, ix->b_row, ix->b_contig, ix->seg, ix->flags);
    else hipLaunchKernelGGL(k_v3_local<false>, dim3(V3_BUCKETS >> ms), dim3(OS_THREADS), local_lds, ctx->stream, (const int4*)recs, (const uint32_t*)hist, nchunks, n, nc, cap, bin_bits, ms,
                            meta, st_local, ix->b_start, ix->ep, ix->b_row, ix->b_contig, ix->seg, ix->flags);
    t_end(ctx);
    HIP_TRY(hipGetLastError());
    *done = true;
    return IVJ_OK;
}

// direct-address table over the starts (lazily, on the sorted index): per-contig geometry, head marks, look-back max-scan,
// 16-byte bin records
int build_tables(ivj_ctx* ctx, ivj_index* ix) {
    const int nc = ix->n_contigs;
    const int64_t n = ix->n;
    if (n > 0 && nc > 0) {
        const int64_t lb_tiles = (ix->bins_len + LB_TILE - 1) / LB_TILE;
        const size_t zb = align_up((size_t)lb_tiles * 8) + align_up(16);
        IVJ_TRY(arena_reserve(ctx, zb + 4096));
        char* z = arena_take<char>(ctx, zb);
        HIP_TRY(hipMemsetAsync(z, 0, zb, ctx->stream));
        HIP_TRY(hipMemsetAsync(ix->bins, 0, (size_t)ix->bins_len * 4, ctx->stream));
        LAUNCH(ctx, "contig_meta", k_contig_meta, grid1d(nc, 256), 256, (const int32_t*)ix->seg, (const int32_t*)ix->b_start, nc, ix->cmeta);
        LAUNCH(ctx, "bins_mark", k_bins_mark, grid1d(n, 256), 256, (const int32_t*)ix->b_start, (const int32_t*)ix->b_contig, n, nc,
               (const int4*)ix->cmeta, ix->bins);
        LAUNCH(ctx, "bins_scan", (k_scan_lb_u32<MaxOp, false>), lb_tiles, OS_THREADS, ix->bins, ix->bins_len, 0u,
               (uint32_t*)(z + align_up((size_t)lb_tiles * 8)), (unsigned long long*)z);
        LAUNCH(ctx, "bins_records", k_bins_records, grid1d(ix->bins_len, 256), 256, (const uint32_t*)ix->bins, ix->bins_len,
               (const int32_t*)ix->b_start, (const int4*)ix->cmeta, nc, ix->brec);
        HIP_TRY(hipGetLastError());
    }
    ix->tables_built = true;
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
        const size_t v3z = (n > 0 && ix3_wanted(ctx, n, opts->n_contigs)) ? v3_plan(n).zero_bytes : 0;      // the balanced build's zeroed words share the fill
        const size_t small = align_up((nc + 2) * 4) + align_up(16) + 3 * align_up((nc + 1) * 32) + v3z;   // seg, flags, cmeta, cmeta_e, cmeta_j (+ ix3_z)
        const size_t flat_bytes = align_up((nn + 1) * 16) + 3 * align_up((size_t)ix->bins_len * 4);   // rec4, lot, tab2 (filled on demand)
        const size_t spl_bytes = align_up((size_t)SL_MAX_BUCKETS * 8) + align_up((size_t)SL_TAB_CONTIGS * 16) +
                                 align_up((size_t)(4 * SL_MAX_BUCKETS + 2 * SL_TAB_CONTIGS) * 4);
        ix->cs_ok = n > 0 && cs_geom(n, opts->n_contigs, opts->slice_rows > 0 ? opts->slice_rows : ctx->sl_env_rows, ix->cs_g);
        const size_t cs_bytes = ix->cs_ok ? cs_index_bytes(ix->cs_g) : 0;
        const size_t hier_bytes = align_up(hier_shape((int64_t)nn).values * 4);
        const size_t need = hier_bytes + cs_bytes + spl_bytes + flat_bytes + 6 * col + align_up(nn * 8) + align_up((nn + 1) * 32) + 2 * align_up((size_t)ix->bins_len * 4) +
                            3 * align_up((size_t)ix->bins_len * 16) + small + 256;
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
        ix->nrec = (int4*)p; p += align_up((nn + 1) * 32);           // 32-byte nearest records
        ix->bins = (uint32_t*)p; p += align_up((size_t)ix->bins_len * 4);
        ix->bins_e = (uint32_t*)p; p += align_up((size_t)ix->bins_len * 4);
        ix->brec = (int4*)p; p += align_up((size_t)ix->bins_len * 16);
        ix->brec_e = (int4*)p; p += align_up((size_t)ix->bins_len * 16);
        ix->crec = (int4*)p; p += align_up((size_t)ix->bins_len * 16);     // 16-byte joint records
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
        ix->cmeta_j = (int4*)p; p += align_up((nc + 1) * 32);
        ix->ix3_z = v3z ? p : nullptr; p += v3z;
        if (ix->cs_ok) { cs_index_carve(ix, p); p += cs_bytes; }
        ix->hier = (int32_t*)p;
        // seg, flags and cmeta start zeroed: an empty index answers every probe with "no rows"
        hipError_t e = hipMemsetAsync(small_base, 0, small, ctx->stream);
        if (e != hipSuccess) return cleanup(fail(IVJ_EHIP, std::string("hipMemsetAsync(index meta): ") + hipGetErrorString(e)));
    }
    if (n > 0) {
        ix->has_tables = !(with_end_order & 2);
        int r = IVJ_OK;
        bool sorted = false;
        const bool try_v3 = ix3_wanted(ctx, n, opts->n_contigs);
        ++ctx->ix3_builds_since_fallback;
        if (try_v3) { r = index_sort_v3(ctx, ix, build, opts, &sorted); if (r != IVJ_OK) return cleanup(r); }
        if (!sorted) r = index_sort_v2(ctx, ix, build, opts);
        if (r != IVJ_OK) return cleanup(r);
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
