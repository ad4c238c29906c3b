// ivjoin.hip -- host driver + C ABI (include/ivjoin.h) of the MI355X interval-join engine.
//
// Replaces, for the range-operation hot path, what the reference reaches through
//   range_operation_frame        /root/reference/src/lib.rs:79-145
//   do_range_operation & co      /root/reference/src/operation.rs:27-350
//   IntervalJoinExec + COITrees  (datafusion-bio-function-ranges v0.11.0, call sites
//                                 /root/reference/src/operation.rs:146-158,253-263,331-340)
// gfx950 only: no CUDA paths, no CPU fallback -- every entry point fails with IVJ_EHIP when
// no device is usable.
#include "../../include/ivjoin.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "probe.hip.h"
#include "fine.hip.h"
#include "radix_sort.hip.h"
#include "scan.hip.h"

using namespace ivj;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(IVJ_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));         \
    } while (0)

#define IVJ_TRY(expr)                 \
    do {                              \
        int _r = (expr);              \
        if (_r != IVJ_OK) return _r;  \
    } while (0)

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0;
};

struct TimingRec {
    const char* name;
    hipEvent_t a, b;
};

}  // namespace

struct ivj_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    Arena arena;
    // state handed from ivj_overlap_count_dev to ivj_overlap_fill_dev
    char* ov_buf = nullptr;
    size_t ov_cap = 0;
    int64_t ov_n = -1;
    const void* ov_probe_start = nullptr;
    const ivj_index* ov_ix = nullptr;
    int32_t ov_filter = -1;
    int32_t* ov_hi = nullptr;
    int32_t* ov_cnt = nullptr;
    long long* ov_tile = nullptr;   // ntiles + 1: tile bases, last = total
    long long* h_total = nullptr;   // pinned
    // one released index slab kept for reuse (bench/streaming loops rebuild the index every call)
    char* ix_cache = nullptr;
    size_t ix_cache_cap = 0;
    int64_t ov_total = 0;
    // bucketed (partitioned) copies of the probe columns + their row ids, when the partition path ran
    bool ov_part = false;
    int32_t *pt_c = nullptr, *pt_s = nullptr, *pt_e = nullptr, *pt_row = nullptr;
    int32_t *pu_c = nullptr, *pu_s = nullptr, *pu_e = nullptr, *pu_row = nullptr;   // second set (two-level bucketing)
    uint32_t* pt_bstart = nullptr;     // PART_BUCKETS + 1 bucket starts of the last one-level partition
    bool part_attr_set = false;
    // timing
    int timing = 0;          // 0 off, 1 probe kernels only, 2 every kernel
    bool t_open = false;
    std::vector<TimingRec> recs;
    std::vector<hipEvent_t> pool;
    size_t pool_used = 0;
};

struct ivj_index {
    ivj_ctx* ctx = nullptr;
    int32_t table_mode = 0;
    int64_t n = 0;
    int32_t n_contigs = 0;
    int32_t* b_start = nullptr;
    int2* ep = nullptr;
    int4* rec4 = nullptr;
    uint32_t* lot = nullptr;
    uint2* tab2 = nullptr;
    int32_t* b_row = nullptr;
    int32_t* b_contig = nullptr;
    int32_t* seg = nullptr;
    int32_t* flags = nullptr;
    int32_t* e_end = nullptr;
    int32_t* e_pos = nullptr;
    int4* cmeta = nullptr;
    uint32_t* bins = nullptr;
    int4* cmeta_e = nullptr;
    uint32_t* bins_e = nullptr;
    int4* brec = nullptr;
    int4* brec_e = nullptr;
    int32_t* pargmax = nullptr;
    int4* nrec = nullptr;
    int4* cmeta_j = nullptr;
    int4* crec = nullptr;
    int64_t bins_len = 0;
    bool has_end_order = false;
    bool has_argmax = false;
    bool has_flat = false;
    bool has_rec4 = false;
    bool has_tables = true;    // false: built for merge / cluster only (with_end_order & 2)     // rec4 is filled on demand (join + materialisation path, flat path)
    char* slab = nullptr;      // single allocation holding every array above
    size_t slab_cap = 0;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int arena_reserve(ivj_ctx* ctx, size_t bytes) {
    Arena& A = ctx->arena;
    A.off = 0;
    if (bytes <= A.cap) return IVJ_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (A.base) HIP_TRY(hipFree(A.base));
    A.base = nullptr; A.cap = 0;
    size_t want = align_up(bytes + bytes / 8, 1 << 20);
    hipError_t e = hipMalloc((void**)&A.base, want);
    if (e != hipSuccess) return fail(IVJ_ENOMEM, "arena hipMalloc(" + std::to_string(want) + "): " + hipGetErrorString(e));
    A.cap = want;
    return IVJ_OK;
}

template <class T>
T* arena_take(ivj_ctx* ctx, size_t count) {
    Arena& A = ctx->arena;
    size_t bytes = align_up(count * sizeof(T));
    if (A.off + bytes > A.cap) return nullptr;   // reserve() sized wrongly: programming error
    T* p = reinterpret_cast<T*>(A.base + A.off);
    A.off += bytes;
    return p;
}

bool is_probe_kernel(const char* name) {
    return !std::strncmp(name, "overlap_", 8) || !std::strncmp(name, "count_overlaps", 14) || !std::strncmp(name, "nearest", 7) ||
           !std::strncmp(name, "materialize", 11) || !std::strncmp(name, "take", 4) || !std::strncmp(name, "coverage", 8) ||
           !std::strncmp(name, "subtract_", 9) || !std::strncmp(name, "cluster_", 8);
}
void t_begin(ivj_ctx* ctx, const char* name) {
    ctx->t_open = false;
    if (!ctx->timing) return;
    if (ctx->timing == 1 && !is_probe_kernel(name)) return;
    if (ctx->pool_used + 2 > ctx->pool.size()) {
        for (int i = 0; i < 64; ++i) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) return; ctx->pool.push_back(ev); }
    }
    TimingRec r{name, ctx->pool[ctx->pool_used], ctx->pool[ctx->pool_used + 1]};
    ctx->pool_used += 2;
    (void)hipEventRecord(r.a, ctx->stream);
    ctx->recs.push_back(r);
    ctx->t_open = true;
}
void t_end(ivj_ctx* ctx) {
    if (!ctx->t_open) return;
    (void)hipEventRecord(ctx->recs.back().b, ctx->stream);
    ctx->t_open = false;
}

#define LAUNCH(ctx, name, kernel, grid, block, ...)                                   \
    do {                                                                              \
        t_begin(ctx, name);                                                           \
        hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3((unsigned)(block)), 0, (ctx)->stream, __VA_ARGS__); \
        t_end(ctx);                                                                   \
    } while (0)

inline unsigned grid1d(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

// device-wide scan: three launches (reduce, partials, apply)
template <class T, class Op, bool INCLUSIVE>
void device_scan(ivj_ctx* ctx, const char* name, const T* in, T* out, int64_t n, T identity, T* partials, T* total_out) {
    const int64_t tiles = scan_num_tiles(n);
    LAUNCH(ctx, name, (k_scan_reduce<T, Op>), tiles, SCAN_THREADS, in, n, identity, partials);
    LAUNCH(ctx, name, (k_scan_partials<T, Op>), 1, SCAN_THREADS, partials, tiles, identity, total_out);
    LAUNCH(ctx, name, (k_scan_apply<T, Op, INCLUSIVE>), tiles, SCAN_THREADS, in, out, n, identity, (const T*)partials);
}

struct SortBufs {
    uint32_t *kA, *vA, *kB, *vB, *hist, *partials;
};

size_t sort_scratch_elems_hist(int64_t n) { return (size_t)RS_RADIX * (size_t)rs_num_blocks(n); }

// LSD passes over `bits` low bits of the keys in (kA,vA); returns true when the result is in (kB,vB).
bool radix_sort_pairs(ivj_ctx* ctx, const SortBufs& sb, int64_t n, int bits) {
    const int nblocks = rs_num_blocks(n);
    uint32_t *kin = sb.kA, *vin = sb.vA, *kout = sb.kB, *vout = sb.vB;
    bool flipped = false;
    for (int shift = 0; shift < bits; shift += 8) {
        LAUNCH(ctx, "rs_hist", k_rs_hist, nblocks, RS_THREADS, (const uint32_t*)kin, n, shift, sb.hist, nblocks);
        device_scan<uint32_t, SumOp, false>(ctx, "rs_scan", sb.hist, sb.hist, (int64_t)RS_RADIX * nblocks, 0u, sb.partials,
                                             (uint32_t*)nullptr);
        LAUNCH(ctx, "rs_scatter", k_rs_scatter, nblocks, RS_THREADS, (const uint32_t*)kin, (const uint32_t*)vin, kout, vout,
               n, shift, (const uint32_t*)sb.hist, nblocks);
        std::swap(kin, kout); std::swap(vin, vout);
        flipped = !flipped;
    }
    return flipped;
}

int bits_for(uint32_t max_value) {
    int b = 0;
    while (b < 32 && (max_value >> b) != 0) ++b;
    return b == 0 ? 1 : b;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_opts(const ivj_opts* o) {
    if (!o) return fail(IVJ_EINVAL, "opts is NULL");
    if (o->filter_op != IVJ_FILTER_WEAK && o->filter_op != IVJ_FILTER_STRICT) return fail(IVJ_EINVAL, "filter_op must be 0 (Weak) or 1 (Strict)");
    if (o->n_contigs < 0) return fail(IVJ_EINVAL, "n_contigs < 0");
    if (o->table_mode < 0 || o->table_mode > 2) return fail(IVJ_EINVAL, "table_mode must be 0 (auto), 1 (records) or 2 (bins)");
    if (o->partition_mode < 0 || o->partition_mode > 5) return fail(IVJ_EINVAL, "partition_mode must be 0 (auto), 1 (256-way), 2 (never), 3 (fine, fused path only), 4 (two-level) or 5 (flat, fused path only)");
    return IVJ_OK;
}
int check_side(const ivj_side* s, const char* what) {
    if (!s) return fail(IVJ_EINVAL, std::string(what) + " is NULL");
    if (s->n < 0) return fail(IVJ_EINVAL, std::string(what) + ".n < 0");
    if (s->n > 0 && (!s->contig || !s->start || !s->end)) return fail(IVJ_EINVAL, std::string(what) + " has a NULL column");
    if (s->n > 0x7fff0000ll) return fail(IVJ_EINVAL, std::string(what) + ".n exceeds the int32 row-index range");
    return IVJ_OK;
}

IndexView view_of(const ivj_index* ix) {
    IndexView v;
    v.b_start = ix->b_start; v.ep = ix->ep; v.b_row = ix->b_row; v.seg = ix->seg;
    v.e_end = ix->e_end; v.e_pos = ix->e_pos; v.flags = ix->flags; v.n_contigs = ix->n_contigs;
    v.cmeta = ix->cmeta; v.brec = ix->brec; v.cmeta_e = ix->cmeta_e; v.brec_e = ix->brec_e; v.pargmax = ix->pargmax; v.nrec = ix->nrec; v.cmeta_j = ix->cmeta_j; v.crec = ix->crec;
    v.bins = ix->bins; v.bins_e = ix->bins_e; v.rec4 = ix->rec4; v.tab2 = ix->tab2;
    // 16-byte bin records once the 4-byte tables + key arrays no longer fit the XCD L2s anyway
    v.use_rec = ix->table_mode == 1 ? 1 : (ix->table_mode == 2 ? 0 : (ix->n >= (1ll << 20) ? 1 : 0));
    return v;
}

size_t sort_scratch_bytes(int64_t n) {
    const size_t hist = sort_scratch_elems_hist(n);
    return 4 * align_up((size_t)n * 4) + align_up(hist * 4) + align_up((size_t)(scan_num_tiles((int64_t)hist) + 1) * 4);
}
void take_sort_bufs(ivj_ctx* ctx, int64_t n, SortBufs& sb) {
    const size_t hist = sort_scratch_elems_hist(n);
    sb.kA = arena_take<uint32_t>(ctx, n); sb.vA = arena_take<uint32_t>(ctx, n);
    sb.kB = arena_take<uint32_t>(ctx, n); sb.vB = arena_take<uint32_t>(ctx, n);
    sb.hist = arena_take<uint32_t>(ctx, hist);
    sb.partials = arena_take<uint32_t>(ctx, scan_num_tiles((int64_t)hist) + 1);
}

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
    ix->ctx = ctx; ix->n = build->n; ix->n_contigs = opts->n_contigs; ix->table_mode = opts->table_mode;
    const int64_t n = build->n;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    auto cleanup = [&](int code) { ivj_index_free(ix); return code; };
    {
        const size_t col = align_up(nn * 4);
        const size_t nc = (size_t)opts->n_contigs;
        ix->bins_len = 2 * (int64_t)nn + 2 * (int64_t)nc + 16;
        const size_t small = align_up((nc + 2) * 4) + align_up(16) + 3 * align_up((nc + 1) * 32);   // seg, flags, cmeta, cmeta_e, cmeta_j
        const size_t flat_bytes = align_up((nn + 1) * 16) + 3 * align_up((size_t)ix->bins_len * 4);   // rec4, lot, tab2 (filled on demand)
        const size_t need = flat_bytes + 6 * col + align_up(nn * 8) + align_up((nn + 1) * 16) + 2 * align_up((size_t)ix->bins_len * 4) +
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
    *out = ix;
    return IVJ_OK;
}

int ensure_ov(ivj_ctx* ctx, int64_t n, int with_part) {      // 0: none, 1: one permuted column set, 2: two
    const int64_t tiles = (n + PROBE_TILE - 1) / PROBE_TILE;
    const size_t col = align_up((size_t)n * 4);
    const size_t need = (size_t)(2 + 4 * with_part) * col + align_up((size_t)(tiles + 2) * 8) +
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
    }
    if (with_part > 1) {
        ctx->pu_c = (int32_t*)p; p += col;
        ctx->pu_s = (int32_t*)p; p += col;
        ctx->pu_e = (int32_t*)p; p += col;
        ctx->pu_row = (int32_t*)p; p += col;
    }
    ctx->pt_bstart = (uint32_t*)p; p += align_up((PART_BUCKETS + 1) * 4);
    ctx->ov_tile = (long long*)p;
    return IVJ_OK;
}

// Probe bucketing pays once the index no longer fits the L2s and there are enough probes to
// amortise the two extra passes.  opts->partition_mode: 0 auto, 1 always, 2 never.
bool want_partition(const ivj_index* ix, int64_t n_probe, const ivj_opts* opts) {
    if (opts->partition_mode == 1 || opts->partition_mode >= 3) return true;
    if (opts->partition_mode == 2) return false;
    return n_probe >= (4ll << 20) && ix->n >= (256ll << 10);
}

// one stable 256-way pass: src columns -> dst columns
int partition_pass(ivj_ctx* ctx, ivj_index* ix, bool strict, const int32_t* sc, const int32_t* ss, const int32_t* se,
                   const int32_t* srow, int64_t n, int packed, int32_t* dc, int32_t* ds, int32_t* de, int32_t* drow) {
    const int ntiles = (int)((n + PART_TILE - 1) / PART_TILE);
    const int grid = 8 * ((ntiles + 7) / 8);
    const size_t hist = (size_t)PART_BUCKETS * (size_t)ntiles;
    IVJ_TRY(arena_reserve(ctx, align_up(hist * 4) + align_up((size_t)(scan_num_tiles((int64_t)hist) + 1) * 4) + 4096));
    uint32_t* blk = arena_take<uint32_t>(ctx, hist);
    uint32_t* partials = arena_take<uint32_t>(ctx, scan_num_tiles((int64_t)hist) + 1);
    if (!ctx->part_attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_part_scatter<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PART_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_part_scatter<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PART_LDS_BYTES));
        ctx->part_attr_set = true;
    }
    IndexView v = view_of(ix);
    const bool hvec = aligned16(sc) && aligned16(se);
    if (strict) LAUNCH(ctx, "part_hist", (k_part_hist<true>), grid, PART_THREADS, v, sc, se, n, packed, blk, ntiles, hvec);
    else LAUNCH(ctx, "part_hist", (k_part_hist<false>), grid, PART_THREADS, v, sc, se, n, packed, blk, ntiles, hvec);
    device_scan<uint32_t, SumOp, false>(ctx, "part_scan", blk, blk, (int64_t)hist, 0u, partials, (uint32_t*)nullptr);
    // bucket b starts at blk[b * ntiles] (bucket-major scan); kept for the inverse permutation (k_unpermute)
    HIP_TRY(hipMemcpy2DAsync(ctx->pt_bstart, 4, blk, (size_t)ntiles * 4, 4, PART_BUCKETS, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(ctx->pt_bstart + PART_BUCKETS), (int)n, 1, ctx->stream));
    t_begin(ctx, "part_scatter");
    if (strict)
        hipLaunchKernelGGL((k_part_scatter<true>), dim3(grid), dim3(PART_THREADS), PART_LDS_BYTES, ctx->stream, v, sc, ss, se, srow, n, packed,
                           (const uint32_t*)blk, ntiles, dc, ds, de, drow);
    else
        hipLaunchKernelGGL((k_part_scatter<false>), dim3(grid), dim3(PART_THREADS), PART_LDS_BYTES, ctx->stream, v, sc, ss, se, srow, n, packed,
                           (const uint32_t*)blk, ntiles, dc, ds, de, drow);
    t_end(ctx);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

// partition_mode 4 (or auto for very large probe sides): two stable passes -> 65536 buckets of ~76
// build rows: the 64 probes of a wavefront then look at the same few cache lines.
bool want_two_level(const ivj_index* ix, int64_t n_probe, const ivj_opts* opts) {
    (void)ix; (void)n_probe;
    return opts->partition_mode == 4;
}

int partition_probes(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts) {
    const int64_t n = probe->n;
    const bool strict = opts->filter_op == IVJ_FILTER_STRICT;
    if (want_two_level(ix, n, opts)) {
        int bs = 0;
        while ((ix->bins_len >> bs) > 65533ll) ++bs;
        IVJ_TRY(partition_pass(ctx, ix, strict, probe->contig, probe->start, probe->end, probe->row_id, n, part_pack(bs, 0, true),
                               ctx->pu_c, ctx->pu_s, ctx->pu_e, ctx->pu_row));
        return partition_pass(ctx, ix, strict, ctx->pu_c, ctx->pu_s, ctx->pu_e, ctx->pu_row, n, part_pack(bs, 8, true),
                              ctx->pt_c, ctx->pt_s, ctx->pt_e, ctx->pt_row);
    }
    int bshift = 0;
    while ((ix->bins_len >> bshift) > (int64_t)(PART_BUCKETS - 3)) ++bshift;
    return partition_pass(ctx, ix, strict, probe->contig, probe->start, probe->end, probe->row_id, n, part_pack(bshift, 0, false),
                          ctx->pt_c, ctx->pt_s, ctx->pt_e, ctx->pt_row);
}

int overlap_count(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int64_t* n_pairs) {
    IVJ_TRY(need_tables(ix));
    const int64_t n = probe->n;
    ctx->ov_n = -1;
    if (n == 0 || ix->n == 0) {
        ctx->ov_n = n; ctx->ov_total = 0; ctx->ov_probe_start = probe->start; ctx->ov_ix = ix; ctx->ov_filter = opts->filter_op;
        *n_pairs = 0;
        return IVJ_OK;
    }
    const bool part = want_partition(ix, n, opts);
    IVJ_TRY(ensure_ov(ctx, n, part ? (want_two_level(ix, n, opts) ? 2 : 1) : 0));
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
    ctx->ov_n = n; ctx->ov_probe_start = probe->start; ctx->ov_ix = ix; ctx->ov_filter = opts->filter_op;
    *n_pairs = ctx->ov_total;
    return IVJ_OK;
}

int overlap_fill(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int32_t* out_p, int32_t* out_b,
                 int64_t capacity) {
    if (ctx->ov_n != probe->n || ctx->ov_probe_start != probe->start || ctx->ov_ix != ix || ctx->ov_filter != opts->filter_op)
        return fail(IVJ_ESTATE, "ivj_overlap_fill_dev must follow ivj_overlap_count_dev with the same index, probe and filter_op");
    if (capacity < ctx->ov_total) return fail(IVJ_ECAPACITY, "output capacity " + std::to_string(capacity) + " < " + std::to_string(ctx->ov_total) + " pairs");
    if (ctx->ov_total == 0) return IVJ_OK;
    if (!out_p || !out_b) return fail(IVJ_EINVAL, "output buffers are NULL");
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

// "fine" single pass: 8192-way bucketing (atomics) + join kernel with LDS-resident index slices.
// Available when a bucket spans at most FINE_SLOTS table slots.
bool fine_available(const ivj_index* ix) { return (ix->bins_len >> FINE_SLOT_BITS) <= (int64_t)(FINE_BUCKETS - 2); }

int overlap_fused_fine(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int32_t* out_p, int32_t* out_b,
                       int64_t capacity, int64_t* n_pairs) {
    const int64_t n = probe->n;
    ctx->ov_n = -1;
    *n_pairs = 0;
    if (n == 0 || ix->n == 0) return IVJ_OK;
    IVJ_TRY(ensure_ov(ctx, n, 1));
    int bshift = 0;
    while ((ix->bins_len >> bshift) > (int64_t)(FINE_BUCKETS - 2)) ++bshift;
    const int64_t jgrid = (n + FINE_TILE - 1) / FINE_TILE + FINE_BUCKETS;      // upper bound on the number of tiles
    IVJ_TRY(arena_reserve(ctx, 4 * align_up((size_t)(FINE_BUCKETS + 1) * 4) + align_up((size_t)FINE_BUCKETS * 8) +
                               align_up((size_t)jgrid * 4) + 4096));
    uint32_t* gcount = arena_take<uint32_t>(ctx, FINE_BUCKETS + 1);
    uint32_t* gstart = arena_take<uint32_t>(ctx, FINE_BUCKETS + 1);
    uint32_t* cursor = arena_take<uint32_t>(ctx, FINE_BUCKETS + 1);
    uint32_t* tprefix = arena_take<uint32_t>(ctx, FINE_BUCKETS + 1);
    int2* brange = arena_take<int2>(ctx, FINE_BUCKETS);
    uint32_t* tbucket = arena_take<uint32_t>(ctx, jgrid);
    int4* prec = reinterpret_cast<int4*>(ctx->pt_c);          // the four permuted columns' space holds the 16-byte records
    unsigned long long* state = (unsigned long long*)ctx->ov_tile;
    HIP_TRY(hipMemsetAsync(gcount, 0, (size_t)(FINE_BUCKETS + 1) * 4, ctx->stream));
    HIP_TRY(hipMemsetAsync(state, 0, 16, ctx->stream));
    IndexView v = view_of(ix);
    const bool strict = opts->filter_op == IVJ_FILTER_STRICT;
    const bool vec = aligned16(probe->contig) && aligned16(probe->start) && aligned16(probe->end);
    const int hgrid = 1024;
    if (strict) LAUNCH(ctx, "fine_hist", (k_fine_hist<true>), hgrid, 1024, v, probe->contig, probe->end, n, bshift, vec, gcount);
    else LAUNCH(ctx, "fine_hist", (k_fine_hist<false>), hgrid, 1024, v, probe->contig, probe->end, n, bshift, vec, gcount);
    LAUNCH(ctx, "fine_offsets", k_fine_offsets, 1, 1024, (const uint32_t*)gcount, gstart, cursor, tprefix);
    LAUNCH(ctx, "fine_tilemap", k_fine_tilemap, grid1d(FINE_BUCKETS, 256), 256, (const uint32_t*)ix->bins, (long long)ix->bins_len, bshift,
           (const uint32_t*)tprefix, brange, tbucket);
    const int64_t sgrid = (n + 8192 - 1) / 8192;
    if (strict) LAUNCH(ctx, "fine_scatter", (k_fine_scatter<true>), sgrid, 1024, v, probe->contig, probe->start, probe->end, probe->row_id, n, bshift, vec,
                       cursor, prec);
    else LAUNCH(ctx, "fine_scatter", (k_fine_scatter<false>), sgrid, 1024, v, probe->contig, probe->start, probe->end, probe->row_id, n, bshift, vec,
                cursor, prec);
    if (strict) LAUNCH(ctx, "overlap_fused_fine", (k_overlap_fused_fine<true>), jgrid, FINE_THREADS, v, (const int4*)prec, (const uint32_t*)gstart,
                       (const uint32_t*)tprefix, (const uint32_t*)tbucket, (const int2*)brange, bshift, (long long)ix->bins_len, (long long)capacity,
                       state, out_p, out_b);
    else LAUNCH(ctx, "overlap_fused_fine", (k_overlap_fused_fine<false>), jgrid, FINE_THREADS, v, (const int4*)prec, (const uint32_t*)gstart,
                (const uint32_t*)tprefix, (const uint32_t*)tbucket, (const int2*)brange, bshift, (long long)ix->bins_len, (long long)capacity,
                state, out_p, out_b);
    HIP_TRY(hipMemcpyAsync(ctx->h_total, state, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipGetLastError());
    *n_pairs = ctx->h_total[0];
    if (ctx->h_total[1] != 0)
        return fail(IVJ_ECAPACITY, "output capacity " + std::to_string(capacity) + " < " + std::to_string(ctx->h_total[0]) + " pairs");
    return IVJ_OK;
}

// single pass: (bucketing +) fused count/fill into a caller buffer of known capacity
int overlap_fused(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int32_t* out_p, int32_t* out_b,
                  int64_t capacity, int64_t* n_pairs) {
    IVJ_TRY(need_tables(ix));
    const int64_t n = probe->n;
    ctx->ov_n = -1;                                   // invalidates a pending count -> fill hand-over
    *n_pairs = 0;
    if (n == 0 || ix->n == 0) return IVJ_OK;
    if (opts->partition_mode == 3 && fine_available(ix)) return overlap_fused_fine(ctx, ix, probe, opts, out_p, out_b, capacity, n_pairs);
    const bool part = want_partition(ix, n, opts);
    IVJ_TRY(ensure_ov(ctx, n, part ? (want_two_level(ix, n, opts) ? 2 : 1) : 0));
    // dense results (the caller expects >= 16 pairs per probe; at ~8 the two kernels tie and the flat one still has
    // to fill its arrays and the end order): the flat kernel spreads every window over the whole
    // workgroup (1.8x the count + dense-fill pair on 37 pairs per probe); sparse ones keep the window-scan kernel
    const bool flat = opts->partition_mode == 5 || (opts->partition_mode == 0 && capacity >= 16 * n && ix->n_contigs > 0);
    if (flat) IVJ_TRY(build_flat(ctx, ix));
    // dense tiles of the flat kernel get their match counts from the end order (two-rank formula) instead of a sweep
    const bool rank_counts = flat && capacity >= 16 * n;
    if (rank_counts) IVJ_TRY(build_end_order(ctx, ix));
    const int64_t tiles = flat ? (n + FLAT_TILE - 1) / FLAT_TILE : (n + PROBE_TILE - 1) / PROBE_TILE;
    unsigned long long* state = (unsigned long long*)ctx->ov_tile;   // [0] cursor, [1] overflow
    const int32_t *qc = probe->contig, *qs = probe->start, *qe = probe->end, *ids = probe->row_id;
    if (part) {
        IVJ_TRY(partition_probes(ctx, ix, probe, opts));
        qc = ctx->pt_c; qs = ctx->pt_s; qe = ctx->pt_e; ids = ctx->pt_row;
    }
    HIP_TRY(hipMemsetAsync(state, 0, 16, ctx->stream));
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
    HIP_TRY(hipMemcpyAsync(ctx->h_total, state, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipGetLastError());
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
    LAUNCH(ctx, "unpermute", k_unpermute, (n + UNP_TILE - 1) / UNP_TILE, UNP_THREADS, (const int32_t*)ctx->pt_row, (const uint32_t*)ctx->pt_bstart, n, cols);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

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

int coverage_core(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int64_t* cov) {
    IVJ_TRY(need_tables(ix));
    const int64_t n = probe->n;
    if (n == 0) return IVJ_OK;
    if (ix->n == 0) { HIP_TRY(hipMemsetAsync(cov, 0, (size_t)n * 8, ctx->stream)); return IVJ_OK; }
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

// left minus the union of the index.  capacity < 0: library-allocated device outputs (host path), otherwise the
// caller's buffers; *n_pieces always receives the total.
int subtract_core(ivj_ctx* ctx, ivj_index* ix, const ivj_side* left, const ivj_opts* opts, int64_t capacity, int32_t** o_row,
                  int32_t** o_start, int32_t** o_end, DevBuf* own, int64_t* n_pieces) {
    IVJ_TRY(need_tables(ix));
    const int64_t n = left->n;
    *n_pieces = 0;
    if (n == 0) return IVJ_OK;
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

// fused join + key-column materialisation (k_overlap_fused_rows); same partitioning as overlap_fused
int overlap_fused_rows(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, const ivj_rows* rows, int64_t* n_pairs) {
    IVJ_TRY(need_tables(ix));
    const int64_t n = probe->n;
    const int64_t capacity = rows->n_pairs;
    ctx->ov_n = -1;
    *n_pairs = 0;
    if (n == 0 || ix->n == 0) return IVJ_OK;
    IVJ_TRY(build_rec4(ctx, ix));
    const bool part = want_partition(ix, n, opts);
    IVJ_TRY(ensure_ov(ctx, n, part ? (want_two_level(ix, n, opts) ? 2 : 1) : 0));
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

int count_overlaps_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int64_t* counts) {
    IVJ_TRY(need_tables(ix));
    const int64_t n = probe->n;
    if (n == 0) return IVJ_OK;
    if (ix->n == 0) { HIP_TRY(hipMemsetAsync(counts, 0, (size_t)n * 8, ctx->stream)); return IVJ_OK; }
    IVJ_TRY(build_end_order(ctx, ix));
    // partition_mode 1 only: bucket the probes by genomic position, count in bucket order into scratch, bring the
    // counts back to probe order with the coalesced inverse permutation.  Not the default: with ONE record gather per
    // probe the bucketing + inverse permutation cost more than the L2 locality buys (100M x 5M: 4.1 ms plain, 5.2 ms
    // bucketed; nearest and coverage, with 3+ gathers per probe, do gain).
    const int32_t *qc = probe->contig, *qs = probe->start, *qe = probe->end;
    long long* o_counts = (long long*)counts;
    const bool bucketed = opts->partition_mode == 1 && ix->n > 0;
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
    constexpr int NT = PROBE_THREADS * PROBE_ITEMS_LAT;
    const int64_t tiles = (n + NT - 1) / NT;
    const bool vec = aligned16(qc) && aligned16(qs) && aligned16(qe);
    IndexView v = view_of(ix);
    if (opts->filter_op == IVJ_FILTER_STRICT)
        LAUNCH(ctx, "count_overlaps", (k_count_overlaps<true, PROBE_ITEMS_LAT>), tiles, PROBE_THREADS, v, qc, qs, qe, n, vec, o_counts);
    else
        LAUNCH(ctx, "count_overlaps", (k_count_overlaps<false, PROBE_ITEMS_LAT>), tiles, PROBE_THREADS, v, qc, qs, qe, n, vec, o_counts);
    HIP_TRY(hipGetLastError());
    if (bucketed) {
        UnpermuteCols uc{{o_counts, nullptr, nullptr}, {counts, nullptr, nullptr}, {8, 0, 0}, 1, nullptr};
        IVJ_TRY(unpermute(ctx, n, uc));
    }
    return IVJ_OK;
}

int nearest_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int32_t* idx, int64_t* dist, int32_t* nf) {
    IVJ_TRY(need_tables(ix));
    const int64_t n = probe->n;
    const int k = opts->nearest_k < 1 ? 1 : opts->nearest_k;
    if (n == 0) return IVJ_OK;
    const bool strict = opts->filter_op == IVJ_FILTER_STRICT;
    const bool k1 = k == 1 && opts->include_overlaps;
    if (k1) IVJ_TRY(build_argmax(ctx, ix));
    else IVJ_TRY(build_end_order(ctx, ix));
    // large probe sides: bucket them by genomic position first (every gather of the kernel then stays in the
    // XCD L2s); the kernels write each result to the probe's original row
    const int32_t *qc = probe->contig, *qs = probe->start, *qe = probe->end, *qrow = nullptr;
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
        if (strict) LAUNCH(ctx, "nearest_k1", (k_nearest_k1<true, PROBE_ITEMS_LAT>), 8 * ((tiles + 7) / 8), PROBE_THREADS, v, qc, qs, qe, n, vec, (const int32_t*)nullptr, o_idx, o_dist, o_nf);
        else LAUNCH(ctx, "nearest_k1", (k_nearest_k1<false, PROBE_ITEMS_LAT>), 8 * ((tiles + 7) / 8), PROBE_THREADS, v, qc, qs, qe, n, vec, (const int32_t*)nullptr, o_idx, o_dist, o_nf);
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
    HIP_TRY(hipMemcpyAsync(c, h->contig, (size_t)h->n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(s, h->start, (size_t)h->n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(en, h->end, (size_t)h->n * 4, hipMemcpyHostToDevice, ctx->stream));
    d.s.contig = c; d.s.start = s; d.s.end = en;
    return IVJ_OK;
}

struct IndexHolder {
    ivj_index* ix = nullptr;
    ~IndexHolder() { if (ix) ivj_index_free(ix); }
};
}  // namespace

// =============================================================================== C ABI

extern "C" {

const char* ivj_last_error(void) { return g_err.c_str(); }
const char* ivj_version(void) { return "ivjoin-hip 0.1 (gfx950)"; }

int ivj_device_count(int* n) {
    if (!n) return fail(IVJ_EINVAL, "n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail(IVJ_EHIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *n = c;
    return IVJ_OK;
}

int ivj_ctx_create(int device, ivj_ctx** out) {
    if (!out) return fail(IVJ_EINVAL, "out is NULL");
    int cnt = 0;
    IVJ_TRY(ivj_device_count(&cnt));
    if (device < 0 || device >= cnt) return fail(IVJ_EHIP, "no usable HIP device " + std::to_string(device) + " (device count " + std::to_string(cnt) + ")");
    HIP_TRY(hipSetDevice(device));
    ivj_ctx* ctx = new ivj_ctx();
    ctx->device = device;
    hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return fail(IVJ_EHIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    ctx->stream = ctx->own_stream;
    e = hipHostMalloc((void**)&ctx->h_total, 64, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipStreamDestroy(ctx->own_stream); delete ctx; return fail(IVJ_EHIP, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
    *out = ctx;
    return IVJ_OK;
}

void ivj_ctx_destroy(ivj_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->arena.base) (void)hipFree(ctx->arena.base);
    if (ctx->ov_buf) (void)hipFree(ctx->ov_buf);
    if (ctx->ix_cache) (void)hipFree(ctx->ix_cache);
    if (ctx->h_total) (void)hipHostFree(ctx->h_total);
    for (hipEvent_t ev : ctx->pool) (void)hipEventDestroy(ev);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int ivj_ctx_set_stream(ivj_ctx* ctx, void* hip_stream) {
    if (!ctx) return fail(IVJ_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stream = (hip_stream == (void*)-1) ? ctx->own_stream : (hipStream_t)hip_stream;
    return IVJ_OK;
}

int ivj_ctx_sync(ivj_ctx* ctx) {
    if (!ctx) return fail(IVJ_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IVJ_OK;
}

int ivj_ctx_enable_timing(ivj_ctx* ctx, int on) {
    if (!ctx) return fail(IVJ_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->timing = on < 0 ? 0 : (on > 2 ? 2 : on);
    ctx->recs.clear();
    ctx->pool_used = 0;
    return IVJ_OK;
}

int ivj_ctx_get_timings(ivj_ctx* ctx, ivj_timing* out, int cap, int* n) {
    if (!ctx || !n) return fail(IVJ_EINVAL, "ctx or n is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<ivj_timing> agg;
    for (const TimingRec& r : ctx->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        size_t j = 0;
        for (; j < agg.size(); ++j) if (std::strcmp(agg[j].name, r.name) == 0) break;
        if (j == agg.size()) {
            ivj_timing t; std::memset(&t, 0, sizeof t);
            std::strncpy(t.name, r.name, sizeof(t.name) - 1);
            agg.push_back(t);
        }
        agg[j].launches += 1; agg[j].ms += ms;
    }
    *n = (int)agg.size();
    for (int i = 0; i < (int)agg.size() && i < cap; ++i) out[i] = agg[i];
    ctx->recs.clear();
    ctx->pool_used = 0;
    return IVJ_OK;
}

// ---------------------------------------------------------------- device-resident API

int ivj_index_build_dev(ivj_ctx* ctx, const ivj_side* build_dev, const ivj_opts* opts, int with_end_order, ivj_index** out) {
    if (!ctx || !out) return fail(IVJ_EINVAL, "ctx or out is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(build_dev, "build"));
    DeviceGuard g(ctx->device);
    return index_build(ctx, build_dev, opts, with_end_order, out);
}

void ivj_index_free(ivj_index* ix) {
    if (!ix) return;
    ivj_ctx* ctx = ix->ctx;
    if (ctx && ctx->ov_ix == ix) { ctx->ov_ix = nullptr; ctx->ov_n = -1; }
    if (ix->slab) {
        if (ctx && ix->slab_cap > ctx->ix_cache_cap) {
            // keep the larger slab for the next index on this context (same stream => ordered reuse)
            char* old = ctx->ix_cache;
            ctx->ix_cache = ix->slab; ctx->ix_cache_cap = ix->slab_cap;
            if (old) { DeviceGuard g(ctx->device); (void)hipFree(old); }
        } else {
            DeviceGuard g(ctx ? ctx->device : 0);
            (void)hipFree(ix->slab);
        }
    }
    delete ix;
}

int ivj_overlap_count_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, int64_t* n_pairs) {
    if (!ctx || !ix || !n_pairs) return fail(IVJ_EINVAL, "ctx, index or n_pairs is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    DeviceGuard g(ctx->device);
    return overlap_count(ctx, ix, probe_dev, opts, n_pairs);
}

int ivj_overlap_fill_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts,
                         int32_t* probe_idx_dev, int32_t* build_idx_dev, int64_t capacity) {
    if (!ctx || !ix) return fail(IVJ_EINVAL, "ctx or index is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    DeviceGuard g(ctx->device);
    return overlap_fill(ctx, ix, probe_dev, opts, probe_idx_dev, build_idx_dev, capacity);
}

int ivj_overlap_fused_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts,
                          int32_t* probe_idx_dev, int32_t* build_idx_dev, int64_t capacity, int64_t* n_pairs) {
    if (!ctx || !ix || !n_pairs) return fail(IVJ_EINVAL, "ctx, index or n_pairs is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    if (capacity < 0 || (capacity > 0 && (!probe_idx_dev || !build_idx_dev))) return fail(IVJ_EINVAL, "bad output buffers");
    DeviceGuard g(ctx->device);
    return overlap_fused(ctx, ix, probe_dev, opts, probe_idx_dev, build_idx_dev, capacity, n_pairs);
}

int ivj_overlap_fused_rows_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, const ivj_rows* rows_dev,
                               int64_t* n_pairs) {
    if (!ctx || !ix || !rows_dev || !n_pairs) return fail(IVJ_EINVAL, "ctx, index, rows or n_pairs is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    if (rows_dev->n_pairs < 0) return fail(IVJ_EINVAL, "capacity (rows->n_pairs) < 0");
    if (opts->partition_mode == 3 || opts->partition_mode == 5) return fail(IVJ_EINVAL, "partition_mode 3 / 5 are not available for the rows path");
    DeviceGuard g(ctx->device);
    return overlap_fused_rows(ctx, ix, probe_dev, opts, rows_dev, n_pairs);
}

int ivj_count_overlaps_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, int64_t* counts_dev) {
    if (!ctx || !ix) return fail(IVJ_EINVAL, "ctx or index is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    if (probe_dev->n > 0 && !counts_dev) return fail(IVJ_EINVAL, "counts is NULL");
    DeviceGuard g(ctx->device);
    return count_overlaps_dev(ctx, ix, probe_dev, opts, counts_dev);
}

int ivj_nearest_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, int32_t* idx_dev,
                    int64_t* dist_dev, int32_t* n_found_dev) {
    if (!ctx || !ix) return fail(IVJ_EINVAL, "ctx or index is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    if (probe_dev->n > 0 && (!idx_dev || !dist_dev || !n_found_dev)) return fail(IVJ_EINVAL, "nearest output buffers are NULL");
    if (opts->nearest_k > 1024) return fail(IVJ_EINVAL, "nearest_k > 1024");
    DeviceGuard g(ctx->device);
    return nearest_dev(ctx, ix, probe_dev, opts, idx_dev, dist_dev, n_found_dev);
}

// ---------------------------------------------------------------- host-buffer API

int ivj_overlap(ivj_ctx* ctx, const ivj_side* probe, const ivj_side* build, const ivj_opts* opts, ivj_pairs* out) {
    if (!ctx || !out) return fail(IVJ_EINVAL, "ctx or out is NULL");
    out->n_pairs = 0; out->probe_idx = nullptr; out->build_idx = nullptr;
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe, "probe"));
    IVJ_TRY(check_side(build, "build"));
    DeviceGuard g(ctx->device);
    DevSide dp, db;
    IVJ_TRY(upload_side(ctx, build, db));
    IVJ_TRY(upload_side(ctx, probe, dp));
    IndexHolder h;
    IVJ_TRY(index_build(ctx, &db.s, opts, 0, &h.ix));
    int64_t total = 0;
    IVJ_TRY(overlap_count(ctx, h.ix, &dp.s, opts, &total));
    if (total == 0) return IVJ_OK;
    DevBuf op, ob;
    hipError_t e = hipMalloc(&op.p, (size_t)total * 4);
    if (e == hipSuccess) e = hipMalloc(&ob.p, (size_t)total * 4);
    if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string("hipMalloc(pairs): ") + hipGetErrorString(e));
    IVJ_TRY(overlap_fill(ctx, h.ix, &dp.s, opts, (int32_t*)op.p, (int32_t*)ob.p, total));
    out->probe_idx = (int32_t*)std::malloc((size_t)total * 4);
    out->build_idx = (int32_t*)std::malloc((size_t)total * 4);
    if (!out->probe_idx || !out->build_idx) { ivj_pairs_free(out); return fail(IVJ_ENOMEM, "host malloc(pairs)"); }
    HIP_TRY(hipMemcpyAsync(out->probe_idx, op.p, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out->build_idx, ob.p, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    out->n_pairs = total;
    return IVJ_OK;
}

// ---------------------------------------------------------------- merge / cluster / coverage

int ivj_cluster_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_opts* opts, int64_t min_dist, int64_t* cluster_dev, int32_t* cluster_start_dev,
                    int32_t* cluster_end_dev, int64_t* n_clusters) {
    if (!ctx || !ix || !n_clusters) return fail(IVJ_EINVAL, "ctx, index or n_clusters is NULL");
    IVJ_TRY(check_opts(opts));
    if (min_dist < 0) return fail(IVJ_EINVAL, "min_dist < 0");
    if (ix->n > 0 && (!cluster_dev || !cluster_start_dev || !cluster_end_dev)) return fail(IVJ_EINVAL, "cluster output buffers are NULL");
    DeviceGuard g(ctx->device);
    Clusters cl;
    IVJ_TRY(cluster_core(ctx, ix, opts->filter_op == IVJ_FILTER_STRICT, (long long)min_dist, 0, cl));
    *n_clusters = cl.n;
    if (ix->n == 0) return IVJ_OK;
    LAUNCH(ctx, "cluster_scatter", k_cluster_scatter, grid1d(ix->n, 256), 256, (const int32_t*)ix->b_row, (const uint32_t*)cl.cid1, (const int32_t*)cl.m_start,
           (const int32_t*)cl.m_end, ix->n, (long long*)cluster_dev, cluster_start_dev, cluster_end_dev);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

int ivj_merge_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_opts* opts, int64_t min_dist, int64_t capacity, int32_t* contig_dev, int32_t* start_dev,
                  int32_t* end_dev, int64_t* n_intervals_dev, int64_t* n_merged) {
    if (!ctx || !ix || !n_merged) return fail(IVJ_EINVAL, "ctx, index or n_merged is NULL");
    IVJ_TRY(check_opts(opts));
    if (min_dist < 0 || capacity < 0) return fail(IVJ_EINVAL, "min_dist or capacity < 0");
    DeviceGuard g(ctx->device);
    Clusters cl;
    IVJ_TRY(cluster_core(ctx, ix, opts->filter_op == IVJ_FILTER_STRICT, (long long)min_dist, 0, cl));
    *n_merged = cl.n;
    if (cl.n == 0) return IVJ_OK;
    if (cl.n > capacity) return fail(IVJ_ECAPACITY, "merge output capacity " + std::to_string(capacity) + " < " + std::to_string(cl.n) + " intervals");
    if (!contig_dev || !start_dev || !end_dev || !n_intervals_dev) return fail(IVJ_EINVAL, "merge output buffers are NULL");
    HIP_TRY(hipMemcpyAsync(contig_dev, cl.m_contig, (size_t)cl.n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(start_dev, cl.m_start, (size_t)cl.n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(end_dev, cl.m_end, (size_t)cl.n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    LAUNCH(ctx, "cluster_counts", k_cluster_counts, grid1d(cl.n, 256), 256, (const int32_t*)cl.m_first, cl.n, (long long*)n_intervals_dev);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

int ivj_coverage_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, int64_t* coverage_dev) {
    if (!ctx || !ix) return fail(IVJ_EINVAL, "ctx or index is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    if (probe_dev->n > 0 && !coverage_dev) return fail(IVJ_EINVAL, "coverage is NULL");
    DeviceGuard g(ctx->device);
    return coverage_core(ctx, ix, probe_dev, opts, coverage_dev);
}

void ivj_merged_free(ivj_merged* m) {
    if (!m) return;
    std::free(m->contig); std::free(m->start); std::free(m->end); std::free(m->n_intervals);
    m->contig = m->start = m->end = nullptr; m->n_intervals = nullptr; m->n = 0;
}

int ivj_merge(ivj_ctx* ctx, const ivj_side* side, const ivj_opts* opts, int64_t min_dist, ivj_merged* out) {
    if (!ctx || !out) return fail(IVJ_EINVAL, "ctx or out is NULL");
    std::memset(out, 0, sizeof(*out));
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(side, "frame"));
    if (min_dist < 0) return fail(IVJ_EINVAL, "min_dist < 0");
    if (side->n == 0) return IVJ_OK;
    DeviceGuard g(ctx->device);
    DevSide ds;
    IVJ_TRY(upload_side(ctx, side, ds));
    IndexHolder h;
    IVJ_TRY(index_build(ctx, &ds.s, opts, 2, &h.ix));      // sweep only: no lookup tables
    Clusters cl;
    IVJ_TRY(cluster_core(ctx, h.ix, opts->filter_op == IVJ_FILTER_STRICT, (long long)min_dist, align_up((size_t)(side->n + 1) * 8), cl));
    long long* cnt = arena_take<long long>(ctx, side->n + 1);
    LAUNCH(ctx, "cluster_counts", k_cluster_counts, grid1d(cl.n, 256), 256, (const int32_t*)cl.m_first, cl.n, cnt);
    out->contig = (int32_t*)std::malloc((size_t)cl.n * 4);
    out->start = (int32_t*)std::malloc((size_t)cl.n * 4);
    out->end = (int32_t*)std::malloc((size_t)cl.n * 4);
    out->n_intervals = (int64_t*)std::malloc((size_t)cl.n * 8);
    if (!out->contig || !out->start || !out->end || !out->n_intervals) { ivj_merged_free(out); return fail(IVJ_ENOMEM, "host malloc(merged)"); }
    hipError_t e = hipMemcpyAsync(out->contig, cl.m_contig, (size_t)cl.n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out->start, cl.m_start, (size_t)cl.n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out->end, cl.m_end, (size_t)cl.n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out->n_intervals, cnt, (size_t)cl.n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ivj_merged_free(out); return fail(IVJ_EHIP, std::string("D2H(merged): ") + hipGetErrorString(e)); }
    out->n = cl.n;
    return IVJ_OK;
}

int ivj_cluster(ivj_ctx* ctx, const ivj_side* side, const ivj_opts* opts, int64_t min_dist, int64_t* cluster, int32_t* cluster_start,
                int32_t* cluster_end, int64_t* n_clusters) {
    if (!ctx) return fail(IVJ_EINVAL, "ctx is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(side, "frame"));
    if (side->row_id) return fail(IVJ_EINVAL, "ivj_cluster reports per input row: row_id must be NULL");
    if (n_clusters) *n_clusters = 0;
    if (side->n == 0) return IVJ_OK;
    if (!cluster || !cluster_start || !cluster_end) return fail(IVJ_EINVAL, "cluster output buffers are NULL");
    DeviceGuard g(ctx->device);
    DevSide ds;
    IVJ_TRY(upload_side(ctx, side, ds));
    IndexHolder h;
    IVJ_TRY(index_build(ctx, &ds.s, opts, 2, &h.ix));      // sweep only: no lookup tables
    DevBuf out;
    const size_t n = (size_t)side->n;
    hipError_t e = hipMalloc(&out.p, align_up(n * 8) + 2 * align_up(n * 4));
    if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string("hipMalloc(cluster): ") + hipGetErrorString(e));
    int64_t* d_c = (int64_t*)out.p;
    int32_t* d_s = (int32_t*)((char*)out.p + align_up(n * 8));
    int32_t* d_e = (int32_t*)((char*)d_s + align_up(n * 4));
    int64_t ncl = 0;
    IVJ_TRY(ivj_cluster_dev(ctx, h.ix, opts, min_dist, d_c, d_s, d_e, &ncl));
    HIP_TRY(hipMemcpyAsync(cluster, d_c, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(cluster_start, d_s, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(cluster_end, d_e, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (n_clusters) *n_clusters = ncl;
    return IVJ_OK;
}

int ivj_coverage(ivj_ctx* ctx, const ivj_side* probe, const ivj_side* build, const ivj_opts* opts, int64_t* coverage) {
    if (!ctx) return fail(IVJ_EINVAL, "ctx is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe, "probe"));
    IVJ_TRY(check_side(build, "build"));
    if (probe->n == 0) return IVJ_OK;
    if (!coverage) return fail(IVJ_EINVAL, "coverage is NULL");
    DeviceGuard g(ctx->device);
    DevSide dp, db;
    IVJ_TRY(upload_side(ctx, build, db));
    IVJ_TRY(upload_side(ctx, probe, dp));
    IndexHolder h;
    IVJ_TRY(index_build(ctx, &db.s, opts, 0, &h.ix));
    DevBuf out;
    hipError_t e = hipMalloc(&out.p, (size_t)probe->n * 8);
    if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string("hipMalloc(coverage): ") + hipGetErrorString(e));
    IVJ_TRY(coverage_core(ctx, h.ix, &dp.s, opts, (int64_t*)out.p));
    HIP_TRY(hipMemcpyAsync(coverage, out.p, (size_t)probe->n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IVJ_OK;
}

// ---------------------------------------------------------------- subtract / complement

int ivj_subtract_dev(ivj_ctx* ctx, ivj_index* right_ix, const ivj_side* left_dev, const ivj_opts* opts, int64_t capacity, int32_t* row_dev,
                     int32_t* start_dev, int32_t* end_dev, int64_t* n_pieces) {
    if (!ctx || !right_ix || !n_pieces) return fail(IVJ_EINVAL, "ctx, index or n_pieces is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(left_dev, "left"));
    if (capacity < 0) return fail(IVJ_EINVAL, "capacity < 0");
    DeviceGuard g(ctx->device);
    return subtract_core(ctx, right_ix, left_dev, opts, capacity, &row_dev, &start_dev, &end_dev, nullptr, n_pieces);
}

void ivj_pieces_free(ivj_pieces* p) {
    if (!p) return;
    std::free(p->row); std::free(p->start); std::free(p->end);
    p->row = p->start = p->end = nullptr; p->n = 0;
}

int ivj_subtract(ivj_ctx* ctx, const ivj_side* left, const ivj_side* right, const ivj_opts* opts, ivj_pieces* out) {
    if (!ctx || !out) return fail(IVJ_EINVAL, "ctx or out is NULL");
    std::memset(out, 0, sizeof(*out));
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(left, "left"));
    IVJ_TRY(check_side(right, "right"));
    if (left->n == 0) return IVJ_OK;
    DeviceGuard g(ctx->device);
    DevSide dl, dr;
    IVJ_TRY(upload_side(ctx, right, dr));
    IVJ_TRY(upload_side(ctx, left, dl));
    IndexHolder h;
    IVJ_TRY(index_build(ctx, &dr.s, opts, 0, &h.ix));
    DevBuf own;
    int32_t *d_row = nullptr, *d_start = nullptr, *d_end = nullptr;
    int64_t total = 0;
    IVJ_TRY(subtract_core(ctx, h.ix, &dl.s, opts, -1, &d_row, &d_start, &d_end, &own, &total));
    if (total == 0) return IVJ_OK;
    out->row = (int32_t*)std::malloc((size_t)total * 4);
    out->start = (int32_t*)std::malloc((size_t)total * 4);
    out->end = (int32_t*)std::malloc((size_t)total * 4);
    if (!out->row || !out->start || !out->end) { ivj_pieces_free(out); return fail(IVJ_ENOMEM, "host malloc(pieces)"); }
    hipError_t e = hipMemcpyAsync(out->row, d_row, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out->start, d_start, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out->end, d_end, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ivj_pieces_free(out); return fail(IVJ_EHIP, std::string("D2H(pieces): ") + hipGetErrorString(e)); }
    out->n = total;
    return IVJ_OK;
}

int ivj_complement(ivj_ctx* ctx, const ivj_side* frame, const ivj_side* view, const ivj_opts* opts, ivj_pieces* out) {
    return ivj_subtract(ctx, view, frame, opts, out);      // the gaps of `frame` inside every view interval
}

// ---------------------------------------------------------------- row materialisation

int ivj_materialize_dev(ivj_ctx* ctx, const ivj_side* probe_dev, const ivj_side* build_dev, const ivj_rows* rows) {
    if (!ctx || !rows) return fail(IVJ_EINVAL, "ctx or rows is NULL");
    IVJ_TRY(check_side(probe_dev, "probe"));
    IVJ_TRY(check_side(build_dev, "build"));
    const int64_t n = rows->n_pairs;
    if (n < 0) return fail(IVJ_EINVAL, "n_pairs < 0");
    if (n == 0) return IVJ_OK;
    if (!rows->probe_idx || !rows->build_idx) return fail(IVJ_EINVAL, "pair index columns are NULL");
    DeviceGuard g(ctx->device);
    const bool vec = aligned16(rows->probe_idx) && aligned16(rows->build_idx) && aligned16(rows->contig) && aligned16(rows->start_1) &&
                     aligned16(rows->end_1) && aligned16(rows->start_2) && aligned16(rows->end_2);
    const int64_t per = (int64_t)MAT_THREADS * MAT_ITEMS;
    LAUNCH(ctx, "materialize_keys", k_materialize_keys, (n + per - 1) / per, MAT_THREADS, probe_dev->contig, probe_dev->start, probe_dev->end,
           build_dev->start, build_dev->end, (const int32_t*)rows->probe_idx, (const int32_t*)rows->build_idx, n, vec, rows->contig,
           rows->start_1, rows->end_1, rows->start_2, rows->end_2);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

int ivj_take_dev(ivj_ctx* ctx, const void* src_dev, int32_t elem_bytes, const int32_t* idx_dev, int64_t n, void* dst_dev,
                 uint64_t* validity_dev) {
    if (!ctx) return fail(IVJ_EINVAL, "ctx is NULL");
    if (elem_bytes != 4 && elem_bytes != 8) return fail(IVJ_EINVAL, "elem_bytes must be 4 or 8");
    if (n < 0) return fail(IVJ_EINVAL, "n < 0");
    if (n == 0) return IVJ_OK;
    if (!src_dev || !idx_dev || !dst_dev) return fail(IVJ_EINVAL, "take: NULL buffer");
    DeviceGuard g(ctx->device);
    if (elem_bytes == 4)
        LAUNCH(ctx, "take", (k_take<uint32_t>), grid1d(n, MAT_THREADS), MAT_THREADS, (const uint32_t*)src_dev, idx_dev, n, (uint32_t*)dst_dev,
               (unsigned long long*)validity_dev);
    else
        LAUNCH(ctx, "take", (k_take<unsigned long long>), grid1d(n, MAT_THREADS), MAT_THREADS, (const unsigned long long*)src_dev, idx_dev, n,
               (unsigned long long*)dst_dev, (unsigned long long*)validity_dev);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

void ivj_rows_free(ivj_rows* r) {
    if (!r) return;
    int32_t** cols[7] = {&r->probe_idx, &r->build_idx, &r->contig, &r->start_1, &r->end_1, &r->start_2, &r->end_2};
    for (auto c : cols) { std::free(*c); *c = nullptr; }
    r->n_pairs = 0;
}

int ivj_overlap_rows(ivj_ctx* ctx, const ivj_side* probe, const ivj_side* build, const ivj_opts* opts, ivj_rows* out) {
    if (!ctx || !out) return fail(IVJ_EINVAL, "ctx or out is NULL");
    std::memset(out, 0, sizeof(*out));
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe, "probe"));
    IVJ_TRY(check_side(build, "build"));
    if (probe->row_id || build->row_id) return fail(IVJ_EINVAL, "ivj_overlap_rows gathers by position: row_id must be NULL");
    DeviceGuard g(ctx->device);
    DevSide dp, db;
    IVJ_TRY(upload_side(ctx, build, db));
    IVJ_TRY(upload_side(ctx, probe, dp));
    IndexHolder h;
    IVJ_TRY(index_build(ctx, &db.s, opts, 0, &h.ix));
    int64_t total = 0;
    IVJ_TRY(overlap_count(ctx, h.ix, &dp.s, opts, &total));
    if (total == 0) return IVJ_OK;
    DevBuf cols;                                            // 7 columns in one allocation
    const size_t col = align_up((size_t)total * 4);
    hipError_t e = hipMalloc(&cols.p, 7 * col);
    if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string("hipMalloc(rows): ") + hipGetErrorString(e));
    ivj_rows d;
    d.n_pairs = total;
    int32_t** dcols[7] = {&d.probe_idx, &d.build_idx, &d.contig, &d.start_1, &d.end_1, &d.start_2, &d.end_2};
    for (int k = 0; k < 7; ++k) *dcols[k] = (int32_t*)((char*)cols.p + k * col);
    IVJ_TRY(overlap_fill(ctx, h.ix, &dp.s, opts, d.probe_idx, d.build_idx, total));
    IVJ_TRY(ivj_materialize_dev(ctx, &dp.s, &db.s, &d));
    int32_t** hcols[7] = {&out->probe_idx, &out->build_idx, &out->contig, &out->start_1, &out->end_1, &out->start_2, &out->end_2};
    for (int k = 0; k < 7; ++k) {
        *hcols[k] = (int32_t*)std::malloc((size_t)total * 4);
        if (!*hcols[k]) { ivj_rows_free(out); return fail(IVJ_ENOMEM, "host malloc(rows)"); }
        hipError_t ce = hipMemcpyAsync(*hcols[k], *dcols[k], (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (ce != hipSuccess) { ivj_rows_free(out); return fail(IVJ_EHIP, std::string("D2H(rows): ") + hipGetErrorString(ce)); }
    }
    hipError_t se = hipStreamSynchronize(ctx->stream);
    if (se != hipSuccess) { ivj_rows_free(out); return fail(IVJ_EHIP, std::string("sync(rows): ") + hipGetErrorString(se)); }
    out->n_pairs = total;
    return IVJ_OK;
}

// Arrow C Data Interface (ABI-stable structs of the Arrow specification).
struct ArrowSchema {
    const char* format; const char* name; const char* metadata; int64_t flags; int64_t n_children;
    struct ArrowSchema** children; struct ArrowSchema* dictionary; void (*release)(struct ArrowSchema*); void* private_data;
};
struct ArrowArray {
    int64_t length; int64_t null_count; int64_t offset; int64_t n_buffers; int64_t n_children; const void** buffers;
    struct ArrowArray** children; struct ArrowArray* dictionary; void (*release)(struct ArrowArray*); void* private_data;
};

namespace {
constexpr int kRowCols = 7;
const char* const kRowNames[kRowCols] = {"probe_idx", "build_idx", "contig", "start_1", "end_1", "start_2", "end_2"};

struct RowsSchemaHolder { ArrowSchema child[kRowCols]; ArrowSchema* ptrs[kRowCols]; };
struct RowsArrayHolder { ArrowArray child[kRowCols]; ArrowArray* ptrs[kRowCols]; const void* cbuf[kRowCols][2]; const void* pbuf[1]; };

void release_child_schema(ArrowSchema* s) { s->release = nullptr; }
void release_child_array(ArrowArray* a) { a->release = nullptr; }
void release_rows_schema(ArrowSchema* s) {
    auto* h = static_cast<RowsSchemaHolder*>(s->private_data);
    for (int k = 0; k < kRowCols; ++k) if (h->child[k].release) h->child[k].release(&h->child[k]);
    delete h;
    s->release = nullptr;
}
void release_rows_array(ArrowArray* a) {
    auto* h = static_cast<RowsArrayHolder*>(a->private_data);
    for (int k = 0; k < kRowCols; ++k) {
        std::free(const_cast<void*>(h->cbuf[k][1]));       // the value buffer this array owns
        if (h->child[k].release) h->child[k].release(&h->child[k]);
    }
    delete h;
    a->release = nullptr;
}
}  // namespace

int ivj_side_from_arrow(const void* array, const void* schema, ivj_side* out) {
    if (!array || !schema || !out) return fail(IVJ_EINVAL, "import: NULL argument");
    const auto* arr = static_cast<const ArrowArray*>(array);
    const auto* sch = static_cast<const ArrowSchema*>(schema);
    if (!sch->format || std::strcmp(sch->format, "+s") != 0) return fail(IVJ_EINVAL, "import: a struct array / record batch is expected");
    if (arr->n_children != sch->n_children) return fail(IVJ_EINVAL, "import: array and schema disagree on the number of children");
    if (arr->null_count > 0) return fail(IVJ_EINVAL, "import: the struct array has null rows");
    const int32_t* cols[3] = {nullptr, nullptr, nullptr};
    const char* names[3] = {"contig", "start", "end"};
    for (int64_t k = 0; k < sch->n_children; ++k) {
        const ArrowSchema* cs = sch->children[k];
        const ArrowArray* ca = arr->children[k];
        if (!cs || !ca || !cs->name) continue;
        for (int j = 0; j < 3; ++j) {
            if (std::strcmp(cs->name, names[j]) != 0) continue;
            if (!cs->format || std::strcmp(cs->format, "i") != 0) return fail(IVJ_EINVAL, std::string("import: column ") + names[j] + " must be int32");
            if (ca->null_count > 0) return fail(IVJ_EINVAL, std::string("import: column ") + names[j] + " contains nulls");
            if (ca->length < arr->offset + arr->length) return fail(IVJ_EINVAL, std::string("import: column ") + names[j] + " is shorter than the struct");
            if (ca->n_buffers < 2 || (!ca->buffers[1] && ca->length > 0)) return fail(IVJ_EINVAL, std::string("import: column ") + names[j] + " has no value buffer");
            cols[j] = static_cast<const int32_t*>(ca->buffers[1]) + ca->offset + arr->offset;
            if (ca->length == 0) cols[j] = nullptr;
        }
    }
    for (int j = 0; j < 3; ++j)
        if (!cols[j] && arr->length > 0) return fail(IVJ_EINVAL, std::string("import: no int32 column named ") + names[j]);
    out->contig = cols[0]; out->start = cols[1]; out->end = cols[2];
    out->n = arr->length;
    out->row_id = nullptr;
    return IVJ_OK;
}

int ivj_rows_export_arrow(ivj_rows* rows, void* out_array, void* out_schema) {
    if (!rows || !out_array || !out_schema) return fail(IVJ_EINVAL, "export: NULL argument");
    auto* arr = static_cast<ArrowArray*>(out_array);
    auto* sch = static_cast<ArrowSchema*>(out_schema);
    const int64_t n = rows->n_pairs;
    int32_t** cols[kRowCols] = {&rows->probe_idx, &rows->build_idx, &rows->contig, &rows->start_1, &rows->end_1, &rows->start_2, &rows->end_2};
    for (int k = 0; k < kRowCols; ++k) {
        if (n > 0 && !*cols[k]) return fail(IVJ_EINVAL, std::string("export: column ") + kRowNames[k] + " is NULL");
        if (!*cols[k]) *cols[k] = (int32_t*)std::calloc(1, 4);   // empty result: consumers still expect a buffer
    }
    auto* sh = new RowsSchemaHolder();
    auto* ah = new RowsArrayHolder();
    for (int k = 0; k < kRowCols; ++k) {
        sh->child[k] = ArrowSchema{"i", kRowNames[k], nullptr, 0, 0, nullptr, nullptr, release_child_schema, nullptr};
        sh->ptrs[k] = &sh->child[k];
        ah->cbuf[k][0] = nullptr;                           // no validity bitmap: no nulls
        ah->cbuf[k][1] = *cols[k];
        ah->child[k] = ArrowArray{n, 0, 0, 2, 0, ah->cbuf[k], nullptr, nullptr, release_child_array, nullptr};
        ah->ptrs[k] = &ah->child[k];
        *cols[k] = nullptr;                                 // ownership moved
    }
    rows->n_pairs = 0;
    ah->pbuf[0] = nullptr;
    *sch = ArrowSchema{"+s", "", nullptr, 0, kRowCols, sh->ptrs, nullptr, release_rows_schema, sh};
    *arr = ArrowArray{n, 0, 0, 1, kRowCols, ah->pbuf, ah->ptrs, nullptr, release_rows_array, ah};
    return IVJ_OK;
}

void ivj_pairs_free(ivj_pairs* p) {
    if (!p) return;
    std::free(p->probe_idx); std::free(p->build_idx);
    p->probe_idx = nullptr; p->build_idx = nullptr; p->n_pairs = 0;
}

int ivj_count_overlaps(ivj_ctx* ctx, const ivj_side* probe, const ivj_side* build, const ivj_opts* opts, int64_t* counts) {
    if (!ctx) return fail(IVJ_EINVAL, "ctx is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe, "probe"));
    IVJ_TRY(check_side(build, "build"));
    if (probe->n == 0) return IVJ_OK;
    if (!counts) return fail(IVJ_EINVAL, "counts is NULL");
    DeviceGuard g(ctx->device);
    DevSide dp, db;
    IVJ_TRY(upload_side(ctx, build, db));
    IVJ_TRY(upload_side(ctx, probe, dp));
    IndexHolder h;
    IVJ_TRY(index_build(ctx, &db.s, opts, 1, &h.ix));
    DevBuf dc;
    hipError_t e = hipMalloc(&dc.p, (size_t)probe->n * 8);
    if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string("hipMalloc(counts): ") + hipGetErrorString(e));
    IVJ_TRY(count_overlaps_dev(ctx, h.ix, &dp.s, opts, (int64_t*)dc.p));
    HIP_TRY(hipMemcpyAsync(counts, dc.p, (size_t)probe->n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IVJ_OK;
}

int ivj_nearest(ivj_ctx* ctx, const ivj_side* probe, const ivj_side* build, const ivj_opts* opts, int32_t* idx, int64_t* dist,
                int32_t* n_found) {
    if (!ctx) return fail(IVJ_EINVAL, "ctx is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe, "probe"));
    IVJ_TRY(check_side(build, "build"));
    if (probe->n == 0) return IVJ_OK;
    if (!idx || !dist || !n_found) return fail(IVJ_EINVAL, "nearest output buffers are NULL");
    const int k = opts->nearest_k < 1 ? 1 : opts->nearest_k;
    if (k > 1024) return fail(IVJ_EINVAL, "nearest_k > 1024");
    DeviceGuard g(ctx->device);
    DevSide dp, db;
    IVJ_TRY(upload_side(ctx, build, db));
    IVJ_TRY(upload_side(ctx, probe, dp));
    IndexHolder h;
    const bool general = !(k == 1 && opts->include_overlaps);
    IVJ_TRY(index_build(ctx, &db.s, opts, general ? 1 : 0, &h.ix));
    const size_t slots = (size_t)probe->n * (size_t)k;
    DevBuf di, dd, dn;
    hipError_t e = hipMalloc(&di.p, slots * 4);
    if (e == hipSuccess) e = hipMalloc(&dd.p, slots * 8);
    if (e == hipSuccess) e = hipMalloc(&dn.p, (size_t)probe->n * 4);
    if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string("hipMalloc(nearest): ") + hipGetErrorString(e));
    IVJ_TRY(nearest_dev(ctx, h.ix, &dp.s, opts, (int32_t*)di.p, (int64_t*)dd.p, (int32_t*)dn.p));
    HIP_TRY(hipMemcpyAsync(idx, di.p, slots * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dist, dd.p, slots * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(n_found, dn.p, (size_t)probe->n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IVJ_OK;
}

// ---------------------------------------------------------------- memory helpers

int ivj_dev_alloc(ivj_ctx* ctx, int64_t bytes, void** out) {
    if (!ctx || !out || bytes < 0) return fail(IVJ_EINVAL, "bad argument");
    DeviceGuard g(ctx->device);
    *out = nullptr;
    if (bytes == 0) return IVJ_OK;
    hipError_t e = hipMalloc(out, (size_t)bytes);
    if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return IVJ_OK;
}
int ivj_dev_free(ivj_ctx* ctx, void* p) {
    if (!ctx) return fail(IVJ_EINVAL, "ctx is NULL");
    if (!p) return IVJ_OK;
    DeviceGuard g(ctx->device);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(p));
    return IVJ_OK;
}
int ivj_memcpy_h2d(ivj_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes) {
    if (!ctx || bytes < 0) return fail(IVJ_EINVAL, "bad argument");
    if (bytes == 0) return IVJ_OK;
    DeviceGuard g(ctx->device);
    HIP_TRY(hipMemcpyAsync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IVJ_OK;
}
int ivj_memcpy_d2h(ivj_ctx* ctx, void* dst_host, const void* src_dev, int64_t bytes) {
    if (!ctx || bytes < 0) return fail(IVJ_EINVAL, "bad argument");
    if (bytes == 0) return IVJ_OK;
    DeviceGuard g(ctx->device);
    HIP_TRY(hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IVJ_OK;
}

}  // extern "C"
