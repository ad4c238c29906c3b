// host_cslice.hip.h -- driver of the contig-aligned slice path (cslice.hip.h): geometry, per-index tables, partition, fused join
// Part of the single translation unit ivjoin.hip (included there, in this order); not a stand-alone header.
#pragma once

namespace {

// Geometry for an index of n rows over nc contigs; false when the path does not apply (the callers keep slice.hip.h /
// the 256-bucket kernels).  Known on the host without looking at the data: the bucket SLOTS are an upper bound on the
// slices (sum over contigs of ceil(n_c / R) <= n / R + min(nc, n)).
// Rows per slice, auto: n / 1024 but not below CS_MIN_ROWS -- a 1 M-row index in 1024-row slices runs config 2's join at 0.35 ms,
// in 3072-row slices at 0.23 ms (fewer, longer workgroups; profiles/r04/policy_sweep.txt); IVJ_CS_MIN_ROWS overrides (sweeps).
constexpr int CS_MIN_ROWS = 3072;
bool cs_geom(int64_t n, int nc, int want_rows, CsGeom& g) {
    if (n <= 0 || nc < 1 || nc > CS_MAX_CONTIGS) return false;
    const int64_t extra = (nc < n ? nc : n) + 1;
    int64_t min_rows = CS_MIN_ROWS;
    if (const char* ev = std::getenv("IVJ_CS_MIN_ROWS")) { const int v = std::atoi(ev); if (v > 0) min_rows = v; }
    int64_t R = want_rows > 0 ? want_rows : std::max<int64_t>((n + 1023) / 1024, min_rows);
    R = (R + 63) / 64 * 64;
    if (n / R + extra > SL_MAX_BUCKETS) {
        if (extra + 8 >= SL_MAX_BUCKETS) return false;
        R = ((n + (SL_MAX_BUCKETS - extra) - 1) / (SL_MAX_BUCKETS - extra) + 63) / 64 * 64;
    }
    if (R > SL_MAX_ROWS) return false;
    g.R = (int)R;
    g.nb = (int)(n / R + extra);
    g.n_contigs = nc;
    g.cps = 0; g.ncells = 0;
    for (int cps = 4; cps >= 1 && !g.ncells; cps >>= 1) {
        const int cells = cps * g.nb + 2 * nc;
        if ((size_t)cs_part_lds(g.nb, cells, nc, 4).total <= 160 * 1024) { g.ncells = cells; g.cps = cps; }
    }
    return g.ncells != 0;
}

// bytes of the per-index arrays of this path (part of the index slab)
size_t cs_index_bytes(const CsGeom& g) {
    return align_up((size_t)(g.nb + 2) * 4) + align_up((size_t)g.nb * 8) + align_up((size_t)CS_MAX_CONTIGS * 16) + align_up((size_t)g.ncells * 4) +
           align_up((size_t)g.nb * (size_t)(2 * g.R + CS_BIN_STRIDE_PAD) * 2) + align_up((size_t)g.nb * 32);
}
void cs_index_carve(ivj_index* ix, char* p) {
    const CsGeom& g = ix->cs_g;
    ix->cs_bound = (int32_t*)p; p += align_up((size_t)(g.nb + 2) * 4);
    ix->cs_spl = (unsigned long long*)p; p += align_up((size_t)g.nb * 8);
    ix->cs_cm = (int4*)p; p += align_up((size_t)CS_MAX_CONTIGS * 16);
    ix->cs_cell = (uint32_t*)p; p += align_up((size_t)g.ncells * 4);
    ix->cs_bins = (unsigned short*)p; p += align_up((size_t)g.nb * (size_t)(2 * g.R + CS_BIN_STRIDE_PAD) * 2);
    ix->cs_smeta = (int4*)p;
}

// The path serves the FUSED single pass wherever the slice path is wanted (host_slice.hip.h::want_slices) and the index has
// its arrays; IVJ_CS=0 switches it off (A/B runs against slice.hip.h).
bool cs_wanted(const ivj_ctx* ctx, const ivj_index* ix, const ivj_opts* opts) {
    if (!ix->cs_ok || ctx->cs_env_off) return false;
    // a caller that pins the rows per slice gets the geometry it asked for
    if (opts->slice_rows > 0 && ((opts->slice_rows + 63) / 64 * 64) != ix->cs_g.R) return false;
    return true;
}

constexpr double CS_FAR_LIMIT = 5e-4;

// The probe sample of a call that finds the index without its slice tables rides in the bins launch (k_cs_bins_sample)
struct CsSampleArgs { bool strict; const int32_t *pc, *ps, *pe; int64_t n; uint32_t* gh; unsigned sgrid; size_t lds; };

int cs_ensure_tables(ivj_ctx* ctx, ivj_index* ix, const CsSampleArgs* sample = nullptr) {
    if (ix->cs_built) return IVJ_OK;
    const CsGeom& g = ix->cs_g;
    // (cs_prep_zero: the fused call that launches the tables clears its state words and sample histogram in this kernel)
    const bool pz = ctx->cs_prep_zero && sample != nullptr;
    LAUNCH(ctx, "cs_prep", k_cs_prep, 1, CS_THREADS, (const int32_t*)ix->seg, (const int32_t*)ix->b_start, g, ix->cs_bound, ix->cs_spl, ix->cs_cm, ix->cs_cell,
           pz ? reinterpret_cast<uint32_t*>(ctx->sl_meta + 4) : (uint32_t*)nullptr, pz ? 16 : 0, pz ? ctx->sl_gh : (uint32_t*)nullptr, pz ? g.nb + 4 : 0);
    const size_t lds = (size_t)4 * g.R + (size_t)2 * (2 * g.R + 8);
    if (sample) {
        const CsTab tab{ix->cs_spl, ix->cs_cm, ix->cs_cell};
        const size_t l2 = std::max(lds, sample->lds);
        t_begin(ctx, "cs_bins_sample");
        if (sample->strict)
            hipLaunchKernelGGL((k_cs_bins_sample<true>), dim3((unsigned)g.nb + sample->sgrid), dim3(CS_THREADS), l2, ctx->stream, g.nb, (const int32_t*)ix->cs_bound, (const int32_t*)ix->b_start,
                               (const int2*)ix->ep, (const int32_t*)ix->b_contig, (const int32_t*)ix->seg, g.R, ix->cs_bins, ix->cs_smeta, ix->flags + 1, tab, g, sample->pc, sample->ps,
                               sample->pe, sample->n, sample->gh);
        else
            hipLaunchKernelGGL((k_cs_bins_sample<false>), dim3((unsigned)g.nb + sample->sgrid), dim3(CS_THREADS), l2, ctx->stream, g.nb, (const int32_t*)ix->cs_bound, (const int32_t*)ix->b_start,
                               (const int2*)ix->ep, (const int32_t*)ix->b_contig, (const int32_t*)ix->seg, g.R, ix->cs_bins, ix->cs_smeta, ix->flags + 1, tab, g, sample->pc, sample->ps,
                               sample->pe, sample->n, sample->gh);
        t_end(ctx);
    } else {
        t_begin(ctx, "cs_bins");
        hipLaunchKernelGGL(k_cs_bins, dim3(g.nb), dim3(CS_THREADS), lds, ctx->stream, (const int32_t*)ix->cs_bound, (const int32_t*)ix->b_start, (const int2*)ix->ep,
                           (const int32_t*)ix->b_contig, (const int32_t*)ix->seg, g.R, ix->cs_bins, ix->cs_smeta, ix->flags + 1);
        t_end(ctx);
    }
    HIP_TRY(hipGetLastError());
    // Which join kernel serves this index: the share of the build rows whose prefix max, CS_WIN rows back, still reaches past
    // their start (k_cs_bins counts them) = the share of positions where a short probe's window would NOT settle inside the
    // branch-free one.  Above CS_FAR_LIMIT (a tail of long rows: genes among exons, a contig-wide row) windows that run on are the
    // rule and k_cs_join walks them over the block maxima; below it (the synthetic configs: 2e-4) they are the rare exception
    // and k_cs_join_plain recounts them row by row -- without the walk in its hot loops and without the maxima being built.
    // Round 5: the 4-byte count travels to the host behind an EVENT, not a stream synchronisation -- the partition of the call is
    // queued right behind it and the host reads the count (cs_resolve_tables, before the join is launched) while the scatter
    // runs: no bubble in the stream (round 4 synchronised here: ~ 20 us per index).  IVJ_CS_WALK = 0 / 1 forces the choice.
    // (host words: in the sampled flow k_cs_regions, queued right behind, stores the count into pinned memory itself -- no copy
    // operation; cs_partition records the event behind that kernel)
    if (!ctx->cs_event) HIP_TRY(hipEventCreateWithFlags(&ctx->cs_event, hipEventDisableTiming));
    if (sample && ctx->hw) ctx->cs_far_hw_seq = ++ctx->hw_seq;
    else {
        ctx->cs_far_hw_seq = 0;
        HIP_TRY(hipMemcpyAsync(ctx->h_total + 5, ix->flags + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipEventRecord(ctx->cs_event, ctx->stream));
    }
    ix->cs_far_pending = true;
    ctx->cs_far_owner = ix;
    ix->cs_built = true;
    return IVJ_OK;
}

// the join kernel of the index (and the block maxima, when it is the walking one): called before the first join / fill launch
int cs_resolve_tables(ivj_ctx* ctx, ivj_index* ix) {
    if (!ix->cs_far_pending) return IVJ_OK;
    bool have = false;
    if (ctx->cs_far_owner == ix) {
        HIP_TRY(wait_event(ctx, ctx->cs_event));
        ctx->cs_far_owner = nullptr;
        if (ctx->cs_far_hw_seq == 0) { ix->cs_far = *reinterpret_cast<const int32_t*>(ctx->h_total + 5); have = true; }
        else if (reinterpret_cast<volatile uint32_t*>(ctx->hw)[5] == ctx->cs_far_hw_seq) { ix->cs_far = (int32_t)reinterpret_cast<volatile uint32_t*>(ctx->hw)[4]; have = true; }
        else ++ctx->hw_misses;                                                 // the word did not arrive: read the count by copy
    }
    if (!have) {
        // another index's tables were launched in between and took the pinned slot: read this index's count again (rare)
        HIP_TRY(hipMemcpyAsync(ctx->h_total + 5, ix->flags + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ix->cs_far = *reinterpret_cast<const int32_t*>(ctx->h_total + 5);
    }
    const double share = ix->n > 0 ? (double)ix->cs_far / (double)ix->n : 0.0;
    ix->cs_walk = ctx->cs_env_walk >= 0 ? ctx->cs_env_walk != 0 : share > CS_FAR_LIMIT;
    ix->cs_far_pending = false;
    if (ix->cs_walk) IVJ_TRY(ensure_hier(ctx, ix));
    return IVJ_OK;
}

// Sampled partition (no histogram pass): the unordered scatter only, and only where the record capacity stays a 32-bit count.
// slack per region: >= one partition tile (an overflowing run is parked at its region's start), + n / (16 nb).
bool cs_sampled_wanted(const ivj_ctx* ctx, int64_t n, bool stable) {
    return !stable && ctx->cs_env_sampled != 0 && !ctx->cs_force_exact && n >= (1ll << 16) && n <= (1ll << 30);
}
uint32_t cs_region_slack(const ivj_ctx* ctx, const CsGeom& g, int64_t n) {
    if (ctx->cs_env_slack > 0) return (uint32_t)((ctx->cs_env_slack + 31) & ~31);        // (tests: below a tile, overflowing runs may leave the region -- only with regions that overflow anyway)
    return (uint32_t)((4 * CS_TILE + n / (16 * (int64_t)(g.nb > 0 ? g.nb : 1)) + 31) & ~31ll);       // (>= the largest partition tile: 16 384 probes, round 6)
}
int64_t cs_record_capacity(const ivj_ctx* ctx, const CsGeom& g, int64_t n) {
    return n + n / 4 + n / 32 + (int64_t)(cs_region_slack(ctx, g, n) + 64) * (g.nb + 2) + 4 * CS_TILE;
}

int cs_plan(const ivj_ctx* ctx, const ivj_index* ix, int64_t n, const ivj_opts* opts, SlicePlan& P, int& wcap) {
    const CsGeom& g = ix->cs_g;
    P.g.nb = g.nb; P.g.R = g.R; P.g.ncells = g.ncells; P.g.cps = g.cps;
    int64_t chunk = ((n + 2047) / 2048 + 2 * CS_TILE - 1) / (2 * CS_TILE) * (2 * CS_TILE);     // whole tiles of either size
    if (chunk < 2 * CS_TILE) chunk = 2 * CS_TILE;
    if (chunk > 16 * CS_TILE) chunk = 16 * CS_TILE;
    P.chunk = (int)chunk;
    P.nchunks = (int)((n + chunk - 1) / chunk);
    P.items = CS_ITEMS;
    const int want_chunk = opts->slice_chunk > 0 ? opts->slice_chunk : ctx->sl_env_chunk;
    int64_t jchunk = want_chunk > 0 ? ((int64_t)want_chunk + CS_TILE - 1) / CS_TILE * CS_TILE
                                    : (n >= (32ll << 20) ? 4 * CS_TILE : (n >= (8ll << 20) ? 2 * CS_TILE : CS_TILE));
    if (want_chunk <= 0 && n >= (32ll << 20)) {
        // Large probe sides (round 5, measured on config 3 on two boxes: join 0.905 / 0.854 / 0.862 / 0.948 / 0.928 / 0.872 / 1.06 ms at
        // 16 / 20 / 24 / 28 / 32 / 40 / 64 Ki probes per workgroup): ~ 20 Ki probes per workgroup, and the AVERAGE bucket cut into equal
        // parts -- a bucket of 80 k probes is four workgroups of 20 k, not five of 16 k with the slice loaded once more.
        const int64_t a = n / (g.nb > 0 ? g.nb : 1);
        const int64_t m = std::max<int64_t>(1, (a + 10240) / 20480);
        const int64_t per = (a + m - 1) / m;
        jchunk = std::min<int64_t>(8 * CS_TILE, std::max<int64_t>(2 * CS_TILE, (per + CS_TILE - 1) / CS_TILE * CS_TILE));
    }
    if (jchunk > 64 * CS_TILE) jchunk = 64 * CS_TILE;
    P.jchunk = (int)jchunk;
    P.gmax = (int)(g.nb + (n + jchunk - 1) / jchunk);
    P.tiles_per_chunk = (int)(jchunk / CS_TILE);
    P.ntiles = (int64_t)P.gmax * P.tiles_per_chunk * CS_WAVES;        // slots of the count -> fill pair: one per (tile, wavefront)
    // partition tiles of 8192 probes where the staging fits the LDS (IVJ_CS_PTILE=4096 pins the small tile: A/B runs)
    P.part_items = ((size_t)cs_part_lds(g.nb, g.ncells, g.n_contigs, 8).total <= 160 * 1024 && ctx->cs_env_ptile != 4096 && n >= (1ll << 20)) ? 8 : 4;
    P.part_lds = (size_t)cs_part_lds(g.nb, g.ncells, g.n_contigs, P.part_items).total;
    // round 6: the sampled scatter of 8-byte records takes 12 288-probe tiles where its staging fits (k_cs_scatter12k; IVJ_CS_PTILE=8192
    // pins the 8192-probe form); the chunks are then whole tiles of all three sizes
    P.part12 = P.part_items == 8 && ctx->cs_env_ptile != 8192 && n >= (8ll << 20) && (size_t)cs_part12_lds(g.nb, g.ncells, g.n_contigs).total <= 160 * 1024;
    // ... and 16 384-probe tiles (6 bytes of staging per probe, copy-out by bucket runs) for sides without row ids from 32 M probes on;
    // IVJ_CS_PTILE=12288 pins the 12 288-probe form
    P.part16 = P.part12 && ctx->cs_env_ptile != 12288 && (n >= (32ll << 20) || ctx->cs_env_ptile == 16384) && (size_t)cs_part12_lds(g.nb, g.ncells, g.n_contigs, 16).total <= 160 * 1024;
    if (P.part12) {
        const int64_t unit = P.part16 ? 12 * CS_TILE : 6 * CS_TILE;            // lcm(8192, 12288[, 16384])
        int64_t c12 = ((n + 2047) / 2048 + unit - 1) / unit * unit;
        if (c12 > 24 * CS_TILE) c12 = 24 * CS_TILE;
        if (ctx->cs_env_pchunks > 0) c12 = std::max<int64_t>(unit, ((n + ctx->cs_env_pchunks - 1) / ctx->cs_env_pchunks + unit - 1) / unit * unit);   // IVJ_CS_PCHUNKS: ~ that many scatter workgroups
        P.chunk = (int)c12;
        P.nchunks = (int)((n + c12 - 1) / c12);
    }
    const size_t lds_cap = 160 * 1024;
    const size_t fixed = (size_t)cs_join_lds(g.R, 0).total;
    if (fixed + 16 * 1024 > lds_cap || P.part_lds > lds_cap) return fail(IVJ_EINVAL, "slice geometry does not fit the LDS");
    wcap = (int)((lds_cap - fixed) / (4 * CS_WAVES)) & ~3;
    P.join_lds = (size_t)cs_join_lds(g.R, wcap).total;
    P.stage = wcap;
    return IVJ_OK;
}

// probe side -> bucket-ordered 12-byte records + chunk table of the join.  stable: the deterministic pair (match-any ranking).
int cs_partition(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, const SlicePlan& P, bool stable) {
    const int64_t n = probe->n;
    const CsGeom& g = ix->cs_g;
    const bool strict = opts->filter_op == IVJ_FILTER_STRICT;
    ctx->sl_sampled = cs_sampled_wanted(ctx, n, stable);
    const bool fuse_sample = ctx->sl_sampled && !ix->cs_built && ctx->cs_env_fuse_sample != 0;     // bins + sample in one launch (IVJ_CS_FUSE_SAMPLE=0: two)
    if (!fuse_sample) IVJ_TRY(cs_ensure_tables(ctx, ix));
    const CsTab tab{ix->cs_spl, ix->cs_cm, ix->cs_cell};
    const bool vec = aligned16(probe->contig) && aligned16(probe->start) && aligned16(probe->end) && (!probe->row_id || aligned16(probe->row_id));
    const size_t hist_lds = (size_t)16 * CS_MAX_CONTIGS + (size_t)8 * g.nb + (size_t)4 * g.ncells + 4 * (g.nb + 1);
    const size_t hist = (size_t)(g.nb + 1) * (size_t)P.nchunks;
    if (!ctx->cs_attr_set) {
        IVJ_TRY(set_dyn_lds(&k_cs_hist<true>, 96 * 1024)); IVJ_TRY(set_dyn_lds(&k_cs_hist<false>, 96 * 1024));
        IVJ_TRY(set_dyn_lds((&k_cs_scatter<true, 4, false>), 160 * 1024)); IVJ_TRY(set_dyn_lds((&k_cs_scatter<false, 4, false>), 160 * 1024));
        IVJ_TRY(set_dyn_lds((&k_cs_scatter<true, 8, false>), 160 * 1024));
        IVJ_TRY(set_dyn_lds(&k_cs_join<true, CS_FUSED>, 160 * 1024)); IVJ_TRY(set_dyn_lds(&k_cs_join<false, CS_FUSED>, 160 * 1024));
        IVJ_TRY(set_dyn_lds(&k_cs_join<true, CS_COUNT>, 160 * 1024)); IVJ_TRY(set_dyn_lds(&k_cs_join<false, CS_COUNT>, 160 * 1024));
        IVJ_TRY(set_dyn_lds(&k_cs_join<true, CS_FILL>, 160 * 1024)); IVJ_TRY(set_dyn_lds(&k_cs_join<false, CS_FILL>, 160 * 1024));
        IVJ_TRY(set_dyn_lds(&k_cs_join_plain<true, CS_FUSED>, 160 * 1024)); IVJ_TRY(set_dyn_lds(&k_cs_join_plain<false, CS_FUSED>, 160 * 1024));
        IVJ_TRY(set_dyn_lds(&k_cs_join_plain<true, CS_COUNT>, 160 * 1024)); IVJ_TRY(set_dyn_lds(&k_cs_join_plain<false, CS_COUNT>, 160 * 1024));
        IVJ_TRY(set_dyn_lds(&k_cs_join_plain<true, CS_FILL>, 160 * 1024)); IVJ_TRY(set_dyn_lds(&k_cs_join_plain<false, CS_FILL>, 160 * 1024));
        IVJ_TRY(set_dyn_lds(&k_cs_scatter_stable<true>, 160 * 1024)); IVJ_TRY(set_dyn_lds(&k_cs_scatter_stable<false>, 160 * 1024));
        IVJ_TRY(set_dyn_lds(&k_cs_fill<true, false, false>, 160 * 1024)); IVJ_TRY(set_dyn_lds(&k_cs_fill<false, false, false>, 160 * 1024));
        IVJ_TRY(set_dyn_lds(&k_cs_fill<true, false, true>, 160 * 1024)); IVJ_TRY(set_dyn_lds(&k_cs_fill<false, false, true>, 160 * 1024));
        IVJ_TRY(set_dyn_lds(&k_cs_fill<true, true, false>, 160 * 1024)); IVJ_TRY(set_dyn_lds(&k_cs_fill<false, true, false>, 160 * 1024));
        ctx->cs_attr_set = true;
    }
    int32_t* rec = reinterpret_cast<int32_t*>(ctx->sl_rec);
    if (ctx->sl_sampled) {
        // region sizes from a 1 / 64 sample of the probe side, one returning atomic per (tile, bucket) run in the scatter, no histogram pass
        if (!ctx->cs_sattr_set) {
            IVJ_TRY(set_dyn_lds(&k_cs_sample_hist<true>, 96 * 1024)); IVJ_TRY(set_dyn_lds(&k_cs_sample_hist<false>, 96 * 1024));
            IVJ_TRY(set_dyn_lds(&k_cs_bins_sample<true>, 96 * 1024)); IVJ_TRY(set_dyn_lds(&k_cs_bins_sample<false>, 96 * 1024));
            IVJ_TRY(set_dyn_lds((&k_cs_scatter<true, 4, true>), 160 * 1024)); IVJ_TRY(set_dyn_lds((&k_cs_scatter<false, 4, true>), 160 * 1024));
            IVJ_TRY(set_dyn_lds((&k_cs_scatter<true, 8, true>), 160 * 1024)); IVJ_TRY(set_dyn_lds((&k_cs_scatter<false, 8, true>), 160 * 1024));
            IVJ_TRY(set_dyn_lds((&k_cs_scatter<true, 4, true, true>), 160 * 1024)); IVJ_TRY(set_dyn_lds((&k_cs_scatter<false, 4, true, true>), 160 * 1024));
            IVJ_TRY(set_dyn_lds((&k_cs_scatter<true, 8, true, true>), 160 * 1024)); IVJ_TRY(set_dyn_lds((&k_cs_scatter<false, 8, true, true>), 160 * 1024));
            IVJ_TRY(set_dyn_lds((&k_cs_scatter12k<true, 12>), 160 * 1024)); IVJ_TRY(set_dyn_lds((&k_cs_scatter12k<false, 12>), 160 * 1024));
            IVJ_TRY(set_dyn_lds((&k_cs_scatter12k<true, 16>), 160 * 1024)); IVJ_TRY(set_dyn_lds((&k_cs_scatter12k<false, 16>), 160 * 1024));
            ctx->cs_sattr_set = true;
        }
        const bool pz = ctx->cs_prep_zero && fuse_sample;                    // k_cs_prep clears the histogram (and the call state)
        if (!pz) HIP_TRY(hipMemsetAsync(ctx->sl_gh, 0, (size_t)(g.nb + 4) * 4, ctx->stream));
        const int64_t n_samp = (n + CS_SRATE - 1) / CS_SRATE;
        const unsigned sgrid = (unsigned)std::min<int64_t>(256, (n_samp + CS_THREADS - 1) / CS_THREADS);
        if (fuse_sample) {
            const CsSampleArgs sa{strict, probe->contig, probe->start, probe->end, n, ctx->sl_gh, sgrid, hist_lds + 16};
            IVJ_TRY(cs_ensure_tables(ctx, ix, &sa));
        } else {
            t_begin(ctx, "cs_sample");
            if (strict) hipLaunchKernelGGL((k_cs_sample_hist<true>), dim3(sgrid), dim3(CS_THREADS), hist_lds + 16, ctx->stream, tab, g, probe->contig, probe->start, probe->end, n, ctx->sl_gh);
            else hipLaunchKernelGGL((k_cs_sample_hist<false>), dim3(sgrid), dim3(CS_THREADS), hist_lds + 16, ctx->stream, tab, g, probe->contig, probe->start, probe->end, n, ctx->sl_gh);
            t_end(ctx);
        }
        // the record format of the call (8-byte records where the sample says they fit) is decided in this kernel, on the device
        // (a context whose calls keep overflowing the 8-byte form -- a probe side with inverted or very long rows between the sampled
        // groups -- stops trying it after CS_REC8_GIVE_UP redos in a row: each redo costs a whole partition + join; a call that would have
        // fitted resets the count only by being tried, so the form is offered again every 64th call)
        const bool sticky12 = ctx->cs_rec8_streak >= CS_REC8_GIVE_UP && (++ctx->cs_rec8_skipped & 63) != 0;
        const int allow8 = (ctx->cs_env_rec8 != 0 && !ctx->cs_force_rec12 && !sticky12) ? 1 : 0;
        ctx->cs_rec8_tried = allow8 != 0;
        const bool far_hw = fuse_sample && ix->cs_far_pending && ctx->cs_far_owner == ix && ctx->cs_far_hw_seq != 0;
        LAUNCH(ctx, "cs_regions", k_cs_regions, 1, CS_THREADS, (const uint32_t*)ctx->sl_gh, g.nb, cs_region_slack(ctx, g, n), allow8, ctx->sl_rstart, ctx->sl_rcur, ctx->sl_meta,
               (const int32_t*)(ix->flags + 1), far_hw ? ctx->hw_dev : (uint32_t*)nullptr, ctx->cs_far_hw_seq);
        if (far_hw) HIP_TRY(hipEventRecord(ctx->cs_event, ctx->stream));
        unsigned long long* state = reinterpret_cast<unsigned long long*>(ctx->sl_meta + 4);
        if (allow8) t_begin(ctx, "cs_scatter");
#define IVJ_CS_SCATTER_S(S, I, R8)                                                                                                      \
    hipLaunchKernelGGL((k_cs_scatter<S, I, true, R8>), dim3(P.nchunks), dim3(CS_THREADS), P.part_lds, ctx->stream, tab, g, probe->contig, probe->start, \
                       probe->end, probe->row_id, n, P.chunk, P.nchunks, vec, (const uint32_t*)ctx->sl_rstart, ctx->sl_rcur, state, (const int32_t*)ctx->sl_meta, rec, ctx->sl_env_ablate)
        // both record forms are queued; the one the device-side format word does not name returns at once (allow8 = 0: only the 12-byte form)
        if (allow8) {
            if (P.part12) {
                const int wide = (P.part16 && !probe->row_id) ? 16 : 12;        // 16 384-probe tiles need the rows implied (row = position)
                const size_t ldsw = (size_t)cs_part12_lds(g.nb, g.ncells, g.n_contigs, wide).total;
#define IVJ_CS_SCATTER_W(S, I)                                                                                                          \
    hipLaunchKernelGGL((k_cs_scatter12k<S, I>), dim3(P.nchunks), dim3(CS_THREADS), ldsw, ctx->stream, tab, g, probe->contig, probe->start, probe->end, probe->row_id, n, \
                       P.chunk, P.nchunks, vec, (const uint32_t*)ctx->sl_rstart, ctx->sl_rcur, state, (const int32_t*)ctx->sl_meta, rec, ctx->sl_env_ablate, ptrace)
                unsigned long long* ptrace = nullptr;                           // IVJ_CS_PTRACE=<file> (diagnosis; tools/ptrace.py): phase stamps of the scatter's tiles
                const char* ptrace_path = std::getenv("IVJ_CS_PTRACE");
                if (ptrace_path && ptrace_path[0]) {
                    HIP_TRY(hipMalloc((void**)&ptrace, (size_t)P.nchunks * 64));
                    HIP_TRY(hipMemsetAsync(ptrace, 0, (size_t)P.nchunks * 64, ctx->stream));
                }
                if (strict) { if (wide == 16) IVJ_CS_SCATTER_W(true, 16); else IVJ_CS_SCATTER_W(true, 12); }
                else { if (wide == 16) IVJ_CS_SCATTER_W(false, 16); else IVJ_CS_SCATTER_W(false, 12); }
#undef IVJ_CS_SCATTER_W
                if (ptrace) {
                    std::vector<unsigned long long> h((size_t)P.nchunks * 8);
                    HIP_TRY(hipStreamSynchronize(ctx->stream));
                    HIP_TRY(hipMemcpy(h.data(), ptrace, h.size() * 8, hipMemcpyDeviceToHost));
                    (void)hipFree(ptrace);
                    if (FILE* f = std::fopen(ptrace_path, "ab")) {
                        const unsigned long long head[4] = {0x50545243ull, (unsigned long long)P.nchunks, (unsigned long long)P.chunk, (unsigned long long)wide};
                        std::fwrite(head, 8, 4, f); std::fwrite(h.data(), 8, h.size(), f); std::fclose(f);
                    }
                }
            }
            else if (strict) { if (P.part_items == 8) IVJ_CS_SCATTER_S(true, 8, true); else IVJ_CS_SCATTER_S(true, 4, true); }
            else { if (P.part_items == 8) IVJ_CS_SCATTER_S(false, 8, true); else IVJ_CS_SCATTER_S(false, 4, true); }
            t_end(ctx);
        }
        t_begin(ctx, allow8 ? "cs_scatter12" : "cs_scatter");
        if (strict) { if (P.part_items == 8) IVJ_CS_SCATTER_S(true, 8, false); else IVJ_CS_SCATTER_S(true, 4, false); }
        else { if (P.part_items == 8) IVJ_CS_SCATTER_S(false, 8, false); else IVJ_CS_SCATTER_S(false, 4, false); }
#undef IVJ_CS_SCATTER_S
        t_end(ctx);
        LAUNCH(ctx, "cs_chunks", k_cs_chunks_sampled, 1, SL_THREADS, (const uint32_t*)ctx->sl_rstart, (const uint32_t*)ctx->sl_rcur, g.nb, P.jchunk, ctx->sl_bstart,
               ctx->sl_bend, ctx->sl_meta, ctx->sl_map);
        HIP_TRY(hipGetLastError());
        return IVJ_OK;
    }
    t_begin(ctx, "cs_hist");
    if (strict) hipLaunchKernelGGL((k_cs_hist<true>), dim3(P.nchunks), dim3(CS_THREADS), hist_lds, ctx->stream, tab, g, probe->contig, probe->end, n, P.chunk, P.nchunks, vec, ctx->sl_blk);
    else hipLaunchKernelGGL((k_cs_hist<false>), dim3(P.nchunks), dim3(CS_THREADS), hist_lds, ctx->stream, tab, g, probe->contig, probe->end, n, P.chunk, P.nchunks, vec, ctx->sl_blk);
    t_end(ctx);
    IVJ_TRY((lb_scan_u32<SumOp, true>(ctx, "cs_scan", ctx->sl_blk, (int64_t)hist, 0u)));
    LAUNCH(ctx, "cs_chunks", k_slice_chunks, 1, SL_THREADS, (const uint32_t*)ctx->sl_blk, P.nchunks, g.nb, n, P.jchunk, ctx->sl_bstart, ctx->sl_meta, ctx->sl_map);
    if (stable) {
        const size_t lds = (size_t)cs_part_s_lds(g.nb, g.ncells, g.n_contigs).total;
        const int nbits = bits_for((uint32_t)g.nb);
        t_begin(ctx, "cs_scatter_stable");
        if (strict) hipLaunchKernelGGL((k_cs_scatter_stable<true>), dim3(P.nchunks), dim3(CS_THREADS), lds, ctx->stream, tab, g, nbits, probe->contig, probe->start, probe->end,
                                       probe->row_id, n, P.chunk, P.nchunks, (const uint32_t*)ctx->sl_blk, rec);
        else hipLaunchKernelGGL((k_cs_scatter_stable<false>), dim3(P.nchunks), dim3(CS_THREADS), lds, ctx->stream, tab, g, nbits, probe->contig, probe->start, probe->end,
                                probe->row_id, n, P.chunk, P.nchunks, (const uint32_t*)ctx->sl_blk, rec);
        t_end(ctx);
        HIP_TRY(hipGetLastError());
        return IVJ_OK;
    }
    t_begin(ctx, "cs_scatter");
#define IVJ_CS_SCATTER(S, I)                                                                                                            \
    hipLaunchKernelGGL((k_cs_scatter<S, I, false>), dim3(P.nchunks), dim3(CS_THREADS), P.part_lds, ctx->stream, tab, g, probe->contig, probe->start, \
                       probe->end, probe->row_id, n, P.chunk, P.nchunks, vec, (const uint32_t*)ctx->sl_blk, (uint32_t*)nullptr, (unsigned long long*)nullptr, (const int32_t*)nullptr, rec, ctx->sl_env_ablate)
    if (strict) { if (P.part_items == 8) IVJ_CS_SCATTER(true, 8); else IVJ_CS_SCATTER(true, 4); }
    else {
        // Weak + histogram-first: the 8192-probe tile form of this variant needs 2 spilled VGPRs (12 bytes of scratch per lane) at the
        // kernel's 128-register ceiling; it takes the 4096-probe tiles instead (the chunks are whole tiles of either size)
        const size_t lds4 = (size_t)cs_part_lds(g.nb, g.ncells, g.n_contigs, 4).total;
        hipLaunchKernelGGL((k_cs_scatter<false, 4, false>), dim3(P.nchunks), dim3(CS_THREADS), lds4, ctx->stream, tab, g, probe->contig, probe->start,
                           probe->end, probe->row_id, n, P.chunk, P.nchunks, vec, (const uint32_t*)ctx->sl_blk, (uint32_t*)nullptr, (unsigned long long*)nullptr, (const int32_t*)nullptr, rec, ctx->sl_env_ablate);
    }
#undef IVJ_CS_SCATTER
    t_end(ctx);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

template <int MODE>
int cs_join_launch(ivj_ctx* ctx, ivj_index* ix, const ivj_opts* opts, const SlicePlan& P, long long capacity, int32_t* out_p, int32_t* out_b) {
    const CsGeom& g = ix->cs_g;
    IVJ_TRY(cs_resolve_tables(ctx, ix));
    CsJoinArgs A;
    A.b_start = ix->b_start; A.ep = ix->ep; A.b_row = ix->b_row; A.bins = ix->cs_bins; A.smeta = ix->cs_smeta; A.hier = view_of(ix).hier;
    A.rec = reinterpret_cast<int32_t*>(ctx->sl_rec); A.bstart = ctx->sl_bstart; A.bend = ctx->sl_sampled ? ctx->sl_bend : nullptr; A.meta = ctx->sl_meta; A.wg_map = ctx->sl_map;
    A.R = g.R; A.jchunk = P.jchunk; A.wcap = P.stage; A.ablate = ctx->sl_env_ablate; A.capacity = capacity;
    A.state = reinterpret_cast<unsigned long long*>(ctx->sl_meta + 4);
    A.wslot = ctx->sl_tile;
    A.cache = (MODE == CS_FUSED || ctx->cs_env_nocache) ? nullptr : ctx->sl_cache;
    A.out_probe = out_p; A.out_build = out_b;
    A.hw = nullptr; A.hw_seq = 0; A.done = reinterpret_cast<uint32_t*>(ctx->sl_meta + 10);
    A.trace = nullptr;
    // persistent workgroups of the plain join (one per CU) draw their items from sl_meta[12..19], cleared with the call's state words;
    // the FILL launch of a pair follows a COUNT launch that used them
    const bool persist = !ix->cs_walk && ctx->cs_env_persist != 0 && ctx->n_cus > 0;   // (k_cs_join keeps one item per workgroup: measured twice -- the run loop costs it 18 - 27 spilled registers, config 2 0.227 -> 0.234 ms before, 0.221 -> 0.221 ms after the peeled tile loop)
    A.cursor = persist ? reinterpret_cast<uint32_t*>(ctx->sl_meta + 12) : nullptr;
    A.pmax = ctx->cs_env_pmax > 0 ? ctx->cs_env_pmax : 4;
    A.pgrain = ctx->cs_env_pgrain > 0 ? ctx->cs_env_pgrain : 64;
    if (persist && MODE == CS_FILL) HIP_TRY(hipMemsetAsync(ctx->sl_meta + 12, 0, 32, ctx->stream));
    if (MODE == CS_FUSED && ctx->cs_env_wgtrace && !ix->cs_walk) {            // diagnosis: one record per workgroup, dumped after the call
        if (ctx->cs_trace_cap < (size_t)P.gmax) {
            if (ctx->cs_trace_buf) HIP_TRY(hipFree(ctx->cs_trace_buf));
            ctx->cs_trace_buf = nullptr; ctx->cs_trace_cap = 0;
            HIP_TRY(hipMalloc((void**)&ctx->cs_trace_buf, (size_t)P.gmax * 48));
            ctx->cs_trace_cap = (size_t)P.gmax;
        }
        HIP_TRY(hipMemsetAsync(ctx->cs_trace_buf, 0, (size_t)P.gmax * 48, ctx->stream));
        A.trace = ctx->cs_trace_buf;
    }
    ctx->cs_fused_hw_seq = 0;
    if (MODE == CS_FUSED && ctx->hw) { A.hw = ctx->hw_dev; A.hw_seq = ctx->cs_fused_hw_seq = ++ctx->hw_seq; }
    const unsigned grid = persist ? (unsigned)std::min(ctx->n_cus, 8 * ((P.gmax + 7) / 8)) : 8u * (unsigned)((P.gmax + 7) / 8);
    t_begin(ctx, MODE == CS_FUSED ? "cs_join_fused" : (MODE == CS_COUNT ? "cs_join_count" : "cs_join_fill"));
    const bool strict = opts->filter_op == IVJ_FILTER_STRICT;
    if (ix->cs_walk) {                                     // a tail of long build rows: windows that run on walk the block maxima
        if (strict) hipLaunchKernelGGL((k_cs_join<true, MODE>), dim3(grid), dim3(CS_THREADS), P.join_lds, ctx->stream, A);
        else hipLaunchKernelGGL((k_cs_join<false, MODE>), dim3(grid), dim3(CS_THREADS), P.join_lds, ctx->stream, A);
    } else {
        if (strict) hipLaunchKernelGGL((k_cs_join_plain<true, MODE>), dim3(grid), dim3(CS_THREADS), P.join_lds, ctx->stream, A);
        else hipLaunchKernelGGL((k_cs_join_plain<false, MODE>), dim3(grid), dim3(CS_THREADS), P.join_lds, ctx->stream, A);
    }
    t_end(ctx);
    HIP_TRY(hipGetLastError());
    if (A.trace) {                                         // (synchronous on purpose: a diagnosis run)
        std::vector<unsigned long long> h((size_t)P.gmax * 6);
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipMemcpy(h.data(), ctx->cs_trace_buf, h.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = std::fopen(ctx->cs_env_wgtrace, "ab")) {
            const unsigned long long head[4] = {0x57475452ull, (unsigned long long)P.gmax, (unsigned long long)P.jchunk, (unsigned long long)g.nb};
            std::fwrite(head, 8, 4, f); std::fwrite(h.data(), 8, h.size(), f); std::fclose(f);
        }
    }
    return IVJ_OK;
}

// A sampled partition whose regions overflowed (state bit 4) left records out: the whole call is redone with the histogram-first
// partition.  The flag lives in the context only for the duration of that second attempt.
struct CsExactScope {
    ivj_ctx* ctx;
    explicit CsExactScope(ivj_ctx* c) : ctx(c) { ctx->cs_force_exact = true; ++ctx->cs_sampled_overflows; }
    ~CsExactScope() { ctx->cs_force_exact = false; }
};
// A probe that did not fit the 8-byte record form the sample had chosen (state bit 8): the call is redone with 12-byte records.
struct CsRec12Scope {
    ivj_ctx* ctx;
    explicit CsRec12Scope(ivj_ctx* c) : ctx(c) { ctx->cs_force_rec12 = true; ++ctx->cs_rec8_overflows; ++ctx->cs_rec8_streak; }
    ~CsRec12Scope() { ctx->cs_force_rec12 = false; }
};

int cs_overlap_fused(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int32_t* out_p, int32_t* out_b,
                     int64_t capacity, int64_t* n_pairs) {
    SlicePlan P;
    int wcap = 0;
    IVJ_TRY(cs_plan(ctx, ix, probe->n, opts, P, wcap));
    IVJ_TRY(ensure_sl(ctx, probe->n, P, cs_sampled_wanted(ctx, probe->n, false) ? cs_record_capacity(ctx, ix->cs_g, probe->n) : 0));
    ctx->sl_plan_valid = false;
    // {pairs, flags, record format (0 = 12-byte records), finished workgroups}: the sampled scatter may set bits 4 / 8.  A call that
    // also launches the index's slice tables has k_cs_prep clear these words and the sample histogram (two fills less in the stream).
    struct PrepZero { ivj_ctx* c; ~PrepZero() { c->cs_prep_zero = false; } } prep_zero_scope{ctx};
    ctx->cs_prep_zero = ctx->hw != nullptr && !ix->cs_built && ctx->cs_env_fuse_sample != 0 && cs_sampled_wanted(ctx, probe->n, false);
    if (!ctx->cs_prep_zero) HIP_TRY(hipMemsetAsync(ctx->sl_meta + 4, 0, 64, ctx->stream));           // state, format, finished workgroups, the join's list cursors
    IVJ_TRY(cs_partition(ctx, ix, probe, opts, P, false));
    if (ctx->sl_env_ablate & (256 | 1024 | 2048)) { *n_pairs = 0; HIP_TRY(hipStreamSynchronize(ctx->stream)); return IVJ_OK; }   // profiling: the records are not usable
    IVJ_TRY(cs_join_launch<CS_FUSED>(ctx, ix, opts, P, (long long)capacity, out_p, out_b));
    // {pairs, flags}: from the host words the last join workgroup wrote -- or by copy (no host words, no workgroup, a protocol timeout)
    bool have_state = false;
    if (ctx->cs_fused_hw_seq != 0) {
        HIP_TRY(wait_stream(ctx, ctx->stream));
        const volatile uint32_t* w = reinterpret_cast<volatile uint32_t*>(ctx->hw);
        if (w[12] == ctx->cs_fused_hw_seq) {
            ctx->h_total[0] = (long long)((unsigned long long)w[8] | ((unsigned long long)w[9] << 32));
            ctx->h_total[1] = (long long)((unsigned long long)w[10] | ((unsigned long long)w[11] << 32));
            have_state = true;
        } else ++ctx->hw_misses;
    }
    if (!have_state) {
        HIP_TRY(hipMemcpyAsync(ctx->h_total, ctx->sl_meta + 4, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(wait_stream(ctx, ctx->stream));
    }
    HIP_TRY(hipGetLastError());
    if (std::getenv("IVJ_DEBUG_REDO")) std::fprintf(stderr, "[ivj] cs_overlap_fused: state flags %lld (sampled %d, exact %d, rec12 %d), pairs %lld\n", (long long)ctx->h_total[1], (int)ctx->sl_sampled, (int)ctx->cs_force_exact, (int)ctx->cs_force_rec12, (long long)ctx->h_total[0]);
    if ((ctx->h_total[1] & 4) && ctx->sl_sampled) {
        CsExactScope redo(ctx);
        return cs_overlap_fused(ctx, ix, probe, opts, out_p, out_b, capacity, n_pairs);
    }
    if ((ctx->h_total[1] & CS_STATE_REC8) && ctx->sl_sampled && !ctx->cs_force_rec12) {
        CsRec12Scope redo(ctx);
        return cs_overlap_fused(ctx, ix, probe, opts, out_p, out_b, capacity, n_pairs);
    }
    if (ctx->cs_rec8_tried && !ctx->cs_force_rec12) ctx->cs_rec8_streak = 0;     // the 8-byte form was offered and no probe overflowed it
    *n_pairs = ctx->h_total[0];
    if (ctx->h_total[1] & 2)          // a bounded wait of the fused tile protocol ran out: the pairs cannot be trusted
        return fail(IVJ_EHIP, "tile protocol timeout in the fused slice join: a workgroup waited for a tile base that never came; the result was discarded");
    if (ctx->h_total[1] != 0)
        return fail(IVJ_ECAPACITY, "output capacity " + std::to_string(capacity) + " < " + std::to_string(ctx->h_total[0]) + " pairs");
    return IVJ_OK;
}

// the deterministic pair: stable partition + per-(tile, wavefront) counts + scan, then the fill at the scanned bases
int cs_overlap_count(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int64_t* n_pairs) {
    SlicePlan P;
    int wcap = 0;
    IVJ_TRY(cs_plan(ctx, ix, probe->n, opts, P, wcap));
    // the stable partition (match-any ranking: 1.39 against 0.88 ms for config 3) only where the caller asks for an output that
    // is identical from run to run (opts->deterministic, IVJ_SLICE_STABLE=1); the pair is exact either way
    const bool stable = opts->deterministic != 0 || ctx->sl_env_stable != 0;
    IVJ_TRY(ensure_sl(ctx, probe->n, P, cs_sampled_wanted(ctx, probe->n, stable) ? cs_record_capacity(ctx, ix->cs_g, probe->n) : 0));
    ctx->sl_plan_valid = false;
    HIP_TRY(hipMemsetAsync(ctx->sl_meta + 4, 0, 64, ctx->stream));           // state, format, finished workgroups, the join's list cursors
    IVJ_TRY(cs_partition(ctx, ix, probe, opts, P, stable));
    HIP_TRY(hipMemsetAsync(ctx->sl_tile, 0, (size_t)(P.ntiles + 2) * 8, ctx->stream));
    IVJ_TRY(cs_join_launch<CS_COUNT>(ctx, ix, opts, P, 0, nullptr, nullptr));
    device_scan<long long, SumOp, false>(ctx, "tile_scan", ctx->sl_tile, ctx->sl_tile, P.ntiles, 0ll, ctx->sl_tpart, ctx->sl_tile + P.ntiles);
    HIP_TRY(hipMemcpyAsync(ctx->h_total, ctx->sl_tile + P.ntiles, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->h_total + 1, ctx->sl_meta + 6, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(wait_stream(ctx, ctx->stream));
    HIP_TRY(hipGetLastError());
    if ((ctx->h_total[1] & 4) && ctx->sl_sampled) {
        CsExactScope redo(ctx);
        return cs_overlap_count(ctx, ix, probe, opts, n_pairs);
    }
    if ((ctx->h_total[1] & CS_STATE_REC8) && ctx->sl_sampled && !ctx->cs_force_rec12) {
        CsRec12Scope redo(ctx);
        return cs_overlap_count(ctx, ix, probe, opts, n_pairs);
    }
    if (ctx->cs_rec8_tried && !ctx->cs_force_rec12) ctx->cs_rec8_streak = 0;
    ctx->sl_plan_valid = true;
    ctx->sl_plan = P;
    *n_pairs = *ctx->h_total;
    return IVJ_OK;
}

// FILL from the words COUNT left per probe record (k_cs_fill): no second matching pass; two workgroups per CU where the build
// rows of a slice + the tile's probe rows + the staging fit 80 KB.  IVJ_CS_NOCACHE=1 keeps the round-3 form (match again).
int cs_fill_launch(ivj_ctx* ctx, ivj_index* ix, const ivj_opts* opts, const SlicePlan& P, int32_t* out_p, int32_t* out_b) {
    const CsGeom& g = ix->cs_g;
    IVJ_TRY(cs_resolve_tables(ctx, ix));
    CsJoinArgs A;
    A.b_start = ix->b_start; A.ep = ix->ep; A.b_row = ix->b_row; A.bins = ix->cs_bins; A.smeta = ix->cs_smeta; A.hier = view_of(ix).hier;
    A.rec = reinterpret_cast<int32_t*>(ctx->sl_rec); A.bstart = ctx->sl_bstart; A.bend = ctx->sl_sampled ? ctx->sl_bend : nullptr; A.meta = ctx->sl_meta; A.wg_map = ctx->sl_map;
    A.R = g.R; A.jchunk = P.jchunk; A.ablate = ctx->sl_env_ablate; A.capacity = 0;
    A.state = reinterpret_cast<unsigned long long*>(ctx->sl_meta + 4);
    A.wslot = ctx->sl_tile; A.cache = ctx->sl_cache;
    A.out_probe = out_p; A.out_build = out_b;
    A.hw = nullptr; A.hw_seq = 0; A.done = nullptr; A.trace = nullptr; A.cursor = nullptr; A.pmax = 1; A.pgrain = 1;
    const size_t fixed = (size_t)cs_fill_lds(g.R, 0).total;
    const size_t half = 80 * 1024, full = 160 * 1024;
    // staging entries per wavefront: a wavefront-tile of 256 probes emits ~2 pairs per probe on the benchmark shapes.  One
    // workgroup per CU with the whole LDS as staging and the next tile's words prefetched is the default; IVJ_CS_FILL_TWO=1 runs
    // two workgroups per CU without the prefetch (measured slower: the kernel is bound by its 1.6 GB of pair stores)
    const bool two = !ix->cs_walk && ctx->cs_env_fill_two != 0 && fixed + 4 * CS_WAVES * 384 <= half;
    int wcap = (int)(((two ? half : full) - fixed) / (4 * CS_WAVES)) & ~3;
    if (wcap > 4096) wcap = 4096;
    A.wcap = wcap;
    const size_t lds = (size_t)cs_fill_lds(g.R, wcap).total;
    const unsigned grid = 8u * (unsigned)((P.gmax + 7) / 8);
    const bool strict = opts->filter_op == IVJ_FILTER_STRICT;
    t_begin(ctx, "cs_fill_cached");
    if (ix->cs_walk) {
        if (strict) hipLaunchKernelGGL((k_cs_fill<true, true, false>), dim3(grid), dim3(CS_THREADS), lds, ctx->stream, A);
        else hipLaunchKernelGGL((k_cs_fill<false, true, false>), dim3(grid), dim3(CS_THREADS), lds, ctx->stream, A);
    } else if (two) {
        if (strict) hipLaunchKernelGGL((k_cs_fill<true, false, true>), dim3(grid), dim3(CS_THREADS), lds, ctx->stream, A);
        else hipLaunchKernelGGL((k_cs_fill<false, false, true>), dim3(grid), dim3(CS_THREADS), lds, ctx->stream, A);
    } else {
        if (strict) hipLaunchKernelGGL((k_cs_fill<true, false, false>), dim3(grid), dim3(CS_THREADS), lds, ctx->stream, A);
        else hipLaunchKernelGGL((k_cs_fill<false, false, false>), dim3(grid), dim3(CS_THREADS), lds, ctx->stream, A);
    }
    t_end(ctx);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

int cs_overlap_fill(ivj_ctx* ctx, ivj_index* ix, const ivj_opts* opts, int32_t* out_p, int32_t* out_b) {
    if (!ctx->sl_plan_valid) return fail(IVJ_ESTATE, "slice fill without a matching count");
    if (ctx->cs_env_nocache) return cs_join_launch<CS_FILL>(ctx, ix, opts, ctx->sl_plan, 0, out_p, out_b);
    return cs_fill_launch(ctx, ix, opts, ctx->sl_plan, out_p, out_b);
}

}  // namespace
