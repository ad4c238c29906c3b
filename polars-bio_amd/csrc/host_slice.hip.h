// host_slice.hip.h -- driver of the slice path (slice.hip.h): geometry, splitters, probe partition, join launches
// Part of the single translation unit ivjoin.hip (included there, in this order); not a stand-alone header.
#pragma once

namespace {

inline int pow2_floor(int x) { int p = 1; while (2 * p <= x) p *= 2; return p; }

// Geometry of the slice path for this index; false when the build side does not fit (more than SL_MAX_BUCKETS
// slices of SL_MAX_ROWS rows) -- the callers then keep the 256-bucket window-scan path.
// opts->slice_rows (0 = auto) pins the rows per slice (tests drive many slices on small inputs with it).
bool slice_geom(const ivj_index* ix, const ivj_opts* opts, SliceGeom& g) {
    const int64_t nbuild = ix->n;
    if (nbuild <= 0) return false;
    const int env_rows = ix->ctx ? ix->ctx->sl_env_rows : 0;
    int64_t R = opts->slice_rows > 0 ? opts->slice_rows : (env_rows > 0 ? env_rows : (nbuild + 1023) / 1024);
    R = (R + 63) / 64 * 64;
    if ((nbuild + R - 1) / R > SL_MAX_BUCKETS) R = ((nbuild + SL_MAX_BUCKETS - 1) / SL_MAX_BUCKETS + 63) / 64 * 64;
    if (R > SL_MAX_ROWS) return false;
    g.R = (int)R;
    g.nb = (int)((nbuild + R - 1) / R);
    g.nbits = bits_for((uint32_t)g.nb);
    g.p2 = pow2_floor(g.nb);
    g.p2r = pow2_floor(g.R);
    // direct-address table over the splitters: four (or, for very many slices, two) cells per splitter; small dictionaries only
    g.ncells = 0;
    if (ix->n_contigs >= 1 && ix->n_contigs <= SL_TAB_CONTIGS && !(ix->ctx && ix->ctx->sl_env_notab)) {
        for (int cps = 4; cps >= 2 && !g.ncells; cps -= 2) {
            const int cells = cps * g.nb + 2 * ix->n_contigs;      // k_slice_tab gives every contig at least two cells
            if ((size_t)slice_part_lds(g.nb, cells).total <= 160 * 1024) { g.ncells = cells; g.cps = cps; }
        }
    }
    return true;
}

// The slice path serves the overlap pair kernels: always with partition_mode 6; in auto mode for the FUSED single pass of
// large inputs (config 3: 3.41 ms per step against 3.65 ms on the 256-bucket path).  The deterministic count -> fill pair
// keeps the 256-bucket path in auto mode (its stable scatter and the second matching pass make the slice pair slower:
// 4.9 against 4.5 ms).  IVJ_SLICE_AUTO=0 / 2 switches the auto choice off / on for both (A/B runs).
bool want_slices(const ivj_index* ix, int64_t n_probe, const ivj_opts* opts, SliceGeom& g, bool fused) {
    if (opts->partition_mode != 0 && opts->partition_mode != 6) return false;
    if (opts->partition_mode == 0) {
        const int mode = ix->ctx ? ix->ctx->sl_env_auto : 1;
        // round 3: the contig-aligned form (cslice.hip.h) also wins the deterministic count -> fill pair (config 3: 3.39 ms
        // against 4.05 ms on 256 buckets); the round-2 slice kernels only ever paid for the fused pass
        const bool cs = ix->cs_ok && !(ix->ctx && ix->ctx->cs_env_off) && opts->slice_rows == 0;
        const bool on = mode == 2 || (mode == 1 && (fused || cs));
        // tools/policy_sweep.py.  Round-2 slice kernels (profiles/r02/policy_sweep.txt): against the 256-bucket window scan they only pay
        // once the sorted build side is far larger than the L2s (5 M rows: -5 % at 30 M probes, -6 % at 100 M; 1-2 M rows: +2..+60 %).
        // The contig-aligned form with slices of >= 3072 rows (round 4, profiles/r04/policy_sweep.txt) wins the whole step wherever
        // bucketing pays at all: 4 M x 256 k rows -13 %, 10 M x 1 M (config 2) -18 %, 30 M x 1 M -25 %, 100 M x 5 M -31 %; with the
        // sampled partition also on the shards of an 8-rank run and their chunks (2 M x 625 k x 3 contigs -16 %, 3 M x 625 k -12 %,
        // 12.5 M x 625 k -27 %; 1 M probes: a tie).
        // (round 6, profiles/r06/policy_sweep_small_build_sides.txt: with the persistent join the contig-aligned form also wins on build sides of
        // 64 k - 256 k rows -- 10 M x 192 k x 1 contig 0.339 -> 0.263 ms, 30 M x 200 k 0.633 -> 0.507, 100 M x 200 k 1.77 -> 1.23; a tie at 2 M x 64 k)
        // ... and from 512 Ki probes on (was 1.5 Mi): 0.5 M x 1 M x 24 0.219 -> 0.187 ms, 1 M x 1 M 0.227 -> 0.199, 1.5 M x 1 M 0.245 -> 0.208
        if (cs) { if (!(on && n_probe >= (1ll << 19) && ix->n >= (64ll << 10))) return false; }
        else if (!(on && n_probe >= (24ll << 20) && ix->n >= (4ll << 20))) return false;
    }
    return slice_geom(ix, opts, g);
}

int ensure_splitters(ivj_ctx* ctx, ivj_index* ix, const SliceGeom& g) {
    if (ix->sl_R == g.R && ix->sl_nb == g.nb && ix->sl_ncells == g.ncells) return IVJ_OK;
    LAUNCH(ctx, "slice_splitters", k_slice_splitters, grid1d(g.nb, 256), 256, (const int32_t*)ix->b_contig, (const int32_t*)ix->b_start, ix->n,
           g.R, g.nb, ix->spl);
    if (g.ncells)
        LAUNCH(ctx, "slice_tab", k_slice_tab, 1, SL_THREADS, (const unsigned long long*)ix->spl, g, g.cps, (const int32_t*)ix->seg,
               (const int32_t*)ix->b_start, ix->n_contigs, ix->sl_cm, ix->sl_cell);
    HIP_TRY(hipGetLastError());
    ix->sl_R = g.R; ix->sl_nb = g.nb; ix->sl_ncells = g.ncells;
    return IVJ_OK;
}

int slice_plan(const ivj_index* ix, int64_t n, const ivj_opts* opts, const SliceGeom& g, int ctx_items, SlicePlan& P) {
    P.g = g;
    int64_t chunk = ((n + 2047) / 2048 + SL_TILE - 1) / SL_TILE * SL_TILE;
    if (chunk < SL_TILE) chunk = SL_TILE;
    if (chunk > 16 * SL_TILE) chunk = 16 * SL_TILE;
    P.chunk = (int)chunk;
    P.nchunks = (int)((n + chunk - 1) / chunk);
    P.items = ctx_items == 2 ? 2 : 4;
    const int64_t jtile = (int64_t)SL_THREADS * P.items;
    const int want_chunk = opts->slice_chunk > 0 ? opts->slice_chunk : (ix->ctx ? ix->ctx->sl_env_chunk : 0);
    int64_t jchunk = want_chunk > 0 ? ((int64_t)want_chunk + jtile - 1) / jtile * jtile
                                           : (n >= (32ll << 20) ? 4 * SL_TILE : (n >= (8ll << 20) ? 2 * SL_TILE : SL_TILE));
    if (jchunk > 64 * SL_TILE) jchunk = 64 * SL_TILE;
    P.jchunk = (int)jchunk;
    P.gmax = (int)(g.nb + (n + jchunk - 1) / jchunk);
    P.tiles_per_chunk = (int)(jchunk / jtile);
    P.ntiles = (int64_t)P.gmax * P.tiles_per_chunk;
    P.lds_seg = ix->n_contigs <= SL_LDS_CONTIGS ? 1 : 0;
    P.part_lds = (size_t)slice_part_lds(g.nb, g.ncells).total;
    P.use_bins = (ix->ctx && ix->ctx->sl_env_nobins) ? 0 : 1;
    const size_t fixed = (size_t)16 * g.R + 4 * SL_PAD + (P.use_bins ? (size_t)2 * (2 * g.R + 8) : 0) + (P.lds_seg ? (size_t)4 * ((ix->n_contigs + 2 + 3) & ~3) : 0) + 8 * (2 * SL_WAVES + 1) + 64;
    const size_t lds_cap = 160 * 1024;
    if (fixed + 8 * 1024 > lds_cap || P.part_lds > lds_cap) return fail(IVJ_EINVAL, "slice geometry does not fit the LDS");
    size_t stage = (lds_cap - fixed) / 8 / SL_THREADS * SL_THREADS;
    if (stage > 12 * 1024) stage = 12 * 1024;
    P.stage = (int)stage;
    P.join_lds = fixed + 8 * stage;
    P.join_lds_count = fixed;
    return IVJ_OK;
}

// scratch of the slice path, owned by the context (kept between count and fill)
// rec_cap: records the bucket-ordered buffer must hold (0: n; the sampled partition of cslice.hip.h reserves regions with slack)
int ensure_sl(ivj_ctx* ctx, int64_t n, const SlicePlan& P, int64_t rec_cap = 0) {
    const size_t hist = (size_t)(P.g.nb + 1) * (size_t)P.nchunks;
    const size_t rc = (size_t)(rec_cap > n ? rec_cap : n);
    const size_t need = align_up(rc * 16) + align_up(rc * 8) + 3 * align_up((size_t)(P.g.nb + 4) * 4) + align_up((size_t)(P.g.nb + 4) * 4 * CS_CUR_STRIDE) + align_up(hist * 4) + align_up((size_t)(scan_num_tiles((int64_t)hist) + 1) * 4) +
                        align_up((size_t)(P.g.nb + 2) * 4) + align_up(64) + align_up((size_t)P.gmax * 8) +
                        align_up((size_t)(P.ntiles + 2) * 8) + align_up((size_t)(scan_num_tiles(P.ntiles) + 2) * 8) + 4096;
    if (need > ctx->sl_cap) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->sl_buf) HIP_TRY(hipFree(ctx->sl_buf));
        ctx->sl_buf = nullptr; ctx->sl_cap = 0;
        // (the scatter's speed depends on where this buffer lands physically: 0.93 .. 1.19 ms on config 3 for the same virtual
        // addresses, tools/scatter_layout_probe*.py; one power-of-two block is not better: 1.07 .. 1.14 ms)
        const size_t want = align_up(need + need / 8, 1 << 20);
        hipError_t e = hipMalloc((void**)&ctx->sl_buf, want);
        if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string("slice scratch hipMalloc: ") + hipGetErrorString(e));
        ctx->sl_cap = want;
        if (std::getenv("IVJ_DEBUG_ALLOC")) std::fprintf(stderr, "[ivj] slice scratch %zu bytes at %p\n", want, (void*)ctx->sl_buf);
    }
    char* p = ctx->sl_buf;
    ctx->sl_rec = (int4*)p; p += align_up(rc * 16);
    ctx->sl_cache = (uint2*)p; p += align_up(rc * 8);          // COUNT -> FILL words of the contig-aligned pair (cslice.hip.h)
    ctx->sl_gh = (uint32_t*)p; p += align_up((size_t)(P.g.nb + 4) * 4);         // sampled partition: sample histogram,
    ctx->sl_rstart = (uint32_t*)p; p += align_up((size_t)(P.g.nb + 4) * 4);     //   region starts,
    ctx->sl_rcur = (uint32_t*)p; p += align_up((size_t)(P.g.nb + 4) * 4 * CS_CUR_STRIDE);   //   region cursors (one per 128-byte line),
    ctx->sl_bend = (uint32_t*)p; p += align_up((size_t)(P.g.nb + 4) * 4);       //   bucket ends
    ctx->sl_blk = (uint32_t*)p; p += align_up(hist * 4);
    ctx->sl_part = (uint32_t*)p; p += align_up((size_t)(scan_num_tiles((int64_t)hist) + 1) * 4);
    ctx->sl_bstart = (uint32_t*)p; p += align_up((size_t)(P.g.nb + 2) * 4);
    ctx->sl_meta = (int32_t*)p; p += align_up(64);                 // [0] join workgroups; bytes 16.. = fused cursor + overflow flag
    ctx->sl_map = (int2*)p; p += align_up((size_t)P.gmax * 8);
    ctx->sl_tile = (long long*)p; p += align_up((size_t)(P.ntiles + 2) * 8);
    ctx->sl_tpart = (long long*)p;
    return IVJ_OK;
}

template <class K>
int set_dyn_lds(K kernel, size_t bytes) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return IVJ_OK;
}

// probe side -> bucket-ordered 16-byte records + chunk table of the join
int slice_partition(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, const SlicePlan& P, bool ordered) {
    const int64_t n = probe->n;
    const bool strict = opts->filter_op == IVJ_FILTER_STRICT;
    IVJ_TRY(ensure_splitters(ctx, ix, P.g));
    const bool vec = aligned16(probe->contig) && aligned16(probe->end);
    const size_t hist_lds = (size_t)16 * SL_TAB_CONTIGS + (size_t)8 * P.g.nb + (size_t)4 * P.g.ncells + 4 * (P.g.nb + 1);
    const SliceTab tab{ix->sl_cm, ix->sl_cell};
    const size_t hist = (size_t)(P.g.nb + 1) * (size_t)P.nchunks;
    t_begin(ctx, "slice_hist");
    if (strict) {
        hipLaunchKernelGGL((k_slice_hist<true>), dim3(P.nchunks), dim3(SL_THREADS), hist_lds, ctx->stream, (const unsigned long long*)ix->spl, tab, P.g,
                           ix->n_contigs, probe->contig, probe->end, n, P.chunk, P.nchunks, vec, ctx->sl_blk);
    } else {
        hipLaunchKernelGGL((k_slice_hist<false>), dim3(P.nchunks), dim3(SL_THREADS), hist_lds, ctx->stream, (const unsigned long long*)ix->spl, tab, P.g,
                           ix->n_contigs, probe->contig, probe->end, n, P.chunk, P.nchunks, vec, ctx->sl_blk);
    }
    t_end(ctx);
    IVJ_TRY((lb_scan_u32<SumOp, true>(ctx, "slice_scan", ctx->sl_blk, (int64_t)hist, 0u)));
    LAUNCH(ctx, "slice_chunks", k_slice_chunks, 1, SL_THREADS, (const uint32_t*)ctx->sl_blk, P.nchunks, P.g.nb, n, P.jchunk, ctx->sl_bstart,
           ctx->sl_meta, ctx->sl_map);
    if (!ordered && !ctx->sl_env_stable) {                            // IVJ_SLICE_STABLE=1 forces the stable scatter (A/B runs)
        // unordered scatter.  IVJ_SLICE_SCATTER_THREADS=512 selects 512-thread workgroups (two per CU instead of one; measured
        // equal: 1.12 vs 1.13 ms, the kernel is bound by its instruction count, not by its barriers).  Chunks are multiples of
        // 4096 probes, so both tilings cover them exactly.
        const int thr = ctx->sl_env_sthreads == 512 ? 512 : 1024;
        const size_t ulds = (size_t)slice_part_u_lds(P.g.nb, P.g.ncells, thr).total;
        t_begin(ctx, "slice_scatter_u");
#define IVJ_LAUNCH_SCATTER_U(S, T)                                                                                                     \
        do {                                                                                                                          \
            IVJ_TRY(set_dyn_lds(&k_slice_scatter_u<S, T>, ulds));                                                                     \
            hipLaunchKernelGGL((k_slice_scatter_u<S, T>), dim3(P.nchunks), dim3(T), ulds, ctx->stream, (const unsigned long long*)ix->spl, tab, P.g, \
                               ix->n_contigs, probe->contig, probe->start, probe->end, probe->row_id, n, P.chunk, P.nchunks,        \
                               (const uint32_t*)ctx->sl_blk, ctx->sl_rec);                                                           \
        } while (0)
        if (strict) { if (thr == 512) IVJ_LAUNCH_SCATTER_U(true, 512); else IVJ_LAUNCH_SCATTER_U(true, 1024); }
        else { if (thr == 512) IVJ_LAUNCH_SCATTER_U(false, 512); else IVJ_LAUNCH_SCATTER_U(false, 1024); }
#undef IVJ_LAUNCH_SCATTER_U
        t_end(ctx);
        HIP_TRY(hipGetLastError());
        return IVJ_OK;
    }
    if (strict) IVJ_TRY(set_dyn_lds(&k_slice_scatter<true>, P.part_lds)); else IVJ_TRY(set_dyn_lds(&k_slice_scatter<false>, P.part_lds));
    t_begin(ctx, "slice_scatter");
    if (strict) {
        hipLaunchKernelGGL((k_slice_scatter<true>), dim3(P.nchunks), dim3(SL_THREADS), P.part_lds, ctx->stream, (const unsigned long long*)ix->spl, tab, P.g,
                           ix->n_contigs, probe->contig, probe->start, probe->end, probe->row_id, n, P.chunk, P.nchunks,
                           (const uint32_t*)ctx->sl_blk, ctx->sl_rec);
    } else {
        hipLaunchKernelGGL((k_slice_scatter<false>), dim3(P.nchunks), dim3(SL_THREADS), P.part_lds, ctx->stream, (const unsigned long long*)ix->spl, tab, P.g,
                           ix->n_contigs, probe->contig, probe->start, probe->end, probe->row_id, n, P.chunk, P.nchunks,
                           (const uint32_t*)ctx->sl_blk, ctx->sl_rec);
    }
    t_end(ctx);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

template <int MODE, int ITEMS>
int slice_join_launch_n(ivj_ctx* ctx, ivj_index* ix, const ivj_opts* opts, const SlicePlan& P, long long capacity, int32_t* out_p, int32_t* out_b) {
    SliceJoinArgs A;
    A.b_start = ix->b_start; A.ep = ix->ep; A.b_row = ix->b_row; A.b_contig = ix->b_contig; A.seg = ix->seg; A.n_contigs = ix->n_contigs;
    A.use_bins = P.use_bins;
    A.ablate = ctx->sl_env_ablate;
    A.rec = ctx->sl_rec; A.bstart = ctx->sl_bstart; A.meta = ctx->sl_meta; A.wg_map = ctx->sl_map;
    A.jchunk = P.jchunk; A.stage = P.stage; A.lds_seg = P.lds_seg; A.capacity = capacity;
    A.tile_tot = ctx->sl_tile; A.state = reinterpret_cast<unsigned long long*>(ctx->sl_meta + 4);
    A.out_probe = out_p; A.out_build = out_b;
    const size_t lds = MODE == SL_COUNT ? P.join_lds_count : P.join_lds;
    const unsigned grid = 8u * (unsigned)((P.gmax + 7) / 8);
    const char* name = MODE == SL_COUNT ? "slice_join_count" : (MODE == SL_FILL ? "slice_join_fill" : "slice_join_fused");
    if (opts->filter_op == IVJ_FILTER_STRICT) {
        IVJ_TRY(set_dyn_lds(&k_slice_join<true, MODE, ITEMS>, lds));
        t_begin(ctx, name);
        hipLaunchKernelGGL((k_slice_join<true, MODE, ITEMS>), dim3(grid), dim3(SL_THREADS), lds, ctx->stream, P.g, ix->n, A);
        t_end(ctx);
    } else {
        IVJ_TRY(set_dyn_lds(&k_slice_join<false, MODE, ITEMS>, lds));
        t_begin(ctx, name);
        hipLaunchKernelGGL((k_slice_join<false, MODE, ITEMS>), dim3(grid), dim3(SL_THREADS), lds, ctx->stream, P.g, ix->n, A);
        t_end(ctx);
    }
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}
template <int MODE>
int slice_join_launch(ivj_ctx* ctx, ivj_index* ix, const ivj_opts* opts, const SlicePlan& P, long long capacity, int32_t* out_p, int32_t* out_b) {
    // (four probes per thread -- IVJ_SLICE_ITEMS=4, a tuning knob -- only for the fused pass: the FILL form of it needs 48 bytes of scratch
    // per lane at the kernel's 128 registers, and the pair's two kernels must agree on the tile; slice_overlap_count plans with two)
    if constexpr (MODE == SL_FUSED) {
        if (P.items == 4) return slice_join_launch_n<MODE, 4>(ctx, ix, opts, P, capacity, out_p, out_b);
    }
    return slice_join_launch_n<MODE, 2>(ctx, ix, opts, P, capacity, out_p, out_b);
}

// count pass of the two-pass pair: partition + join<COUNT> + scan of the tile totals
int slice_overlap_count(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, const SliceGeom& g, int64_t* n_pairs) {
    SlicePlan P;
    IVJ_TRY(slice_plan(ix, probe->n, opts, g, 2, P));                            // (two probes per thread for the pair: see slice_join_launch)
    IVJ_TRY(ensure_sl(ctx, probe->n, P));
    IVJ_TRY(slice_partition(ctx, ix, probe, opts, P, true));
    HIP_TRY(hipMemsetAsync(ctx->sl_tile, 0, (size_t)(P.ntiles + 2) * 8, ctx->stream));
    IVJ_TRY(slice_join_launch<SL_COUNT>(ctx, ix, opts, P, 0, nullptr, nullptr));
    device_scan<long long, SumOp, false>(ctx, "tile_scan", ctx->sl_tile, ctx->sl_tile, P.ntiles, 0ll, ctx->sl_tpart, ctx->sl_tile + P.ntiles);
    HIP_TRY(hipMemcpyAsync(ctx->h_total, ctx->sl_tile + P.ntiles, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipGetLastError());
    ctx->sl_plan_valid = true;
    ctx->sl_plan = P;
    *n_pairs = *ctx->h_total;
    return IVJ_OK;
}

int slice_overlap_fill(ivj_ctx* ctx, ivj_index* ix, const ivj_opts* opts, int32_t* out_p, int32_t* out_b) {
    if (!ctx->sl_plan_valid) return fail(IVJ_ESTATE, "slice fill without a matching count");
    return slice_join_launch<SL_FILL>(ctx, ix, opts, ctx->sl_plan, 0, out_p, out_b);
}

int slice_overlap_fused(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, const SliceGeom& g, int32_t* out_p,
                        int32_t* out_b, int64_t capacity, int64_t* n_pairs) {
    SlicePlan P;
    IVJ_TRY(slice_plan(ix, probe->n, opts, g, ctx->sl_items, P));
    IVJ_TRY(ensure_sl(ctx, probe->n, P));
    ctx->sl_plan_valid = false;
    IVJ_TRY(slice_partition(ctx, ix, probe, opts, P, false));
    HIP_TRY(hipMemsetAsync(ctx->sl_meta + 4, 0, 16, ctx->stream));
    IVJ_TRY(slice_join_launch<SL_FUSED>(ctx, ix, opts, P, (long long)capacity, out_p, out_b));
    HIP_TRY(hipMemcpyAsync(ctx->h_total, ctx->sl_meta + 4, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipGetLastError());
    *n_pairs = ctx->h_total[0];
    if (ctx->h_total[1] & 2)          // a bounded wait of the fused tile protocol ran out: the pairs cannot be trusted
        return fail(IVJ_EHIP, "tile protocol timeout in the fused slice join: a workgroup waited for a tile base that never came; the result was discarded");
    if (ctx->h_total[1] != 0)
        return fail(IVJ_ECAPACITY, "output capacity " + std::to_string(capacity) + " < " + std::to_string(ctx->h_total[0]) + " pairs");
    return IVJ_OK;
}

}  // namespace
