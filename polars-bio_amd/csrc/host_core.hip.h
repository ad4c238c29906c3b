// host_core.hip.h -- error state, scratch arena, context and index structs, timing, launch / scan / sort helpers of the host driver
// Part of the single translation unit ivjoin.hip (included there, in this order); not a stand-alone header.
#pragma once

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(IVJ_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));         \
    } while (0)

// Every int-returning entry point of the C ABI is a function-try-block ending in this: no C++ exception (std::bad_alloc of a
// std::vector / std::string, std::system_error of a std::thread) may cross the boundary into a C, Rust or ctypes caller.
#define IVJ_ABI_CATCH                                                                                     \
    catch (const std::bad_alloc&) { return fail(IVJ_ENOMEM, "host allocation failed (std::bad_alloc)"); }   \
    catch (const std::exception& e) { return fail(IVJ_ESTATE, std::string("internal error: ") + e.what()); } \
    catch (...) { return fail(IVJ_ESTATE, "internal error: unknown exception"); }

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

struct SlicePlan {                     // slice path: launch geometry of one call (host_slice.hip.h fills it)
    SliceGeom g{0, 0, 0, 0, 0, 0, 0};
    int chunk = 0, nchunks = 0;        // partition: probes per workgroup, workgroups
    int jchunk = 0, gmax = 0;          // join: probes per workgroup, upper bound on the workgroups
    int tiles_per_chunk = 0;
    int64_t ntiles = 0;                // gmax * tiles_per_chunk
    int stage = 0, lds_seg = 0, items = 4, use_bins = 1, part_items = 4;
    bool part12 = false;               // the sampled scatter of 8-byte records on 12 288-probe tiles (k_cs_scatter12k)
    bool part16 = false;               // ... on 16 384-probe tiles where the side brings no row ids
    size_t part_lds = 0, join_lds = 0, join_lds_count = 0;
};

// pinned staging slots of the host <-> HBM copies (host_mem.hip.h: HostXfer)
struct XferSlots {                                         // owned by the context, allocated on first use
    char* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    void release() {
        for (int k = 0; k < 2; ++k) {
            if (buf[k]) (void)hipHostFree(buf[k]);
            if (ev[k]) (void)hipEventDestroy(ev[k]);
            buf[k] = nullptr; ev[k] = nullptr;
        }
    }
};

inline std::atomic<int> g_live_contexts{0};     // contexts alive in this process (the hot-path waits yield when there are several)

struct ivj_ctx {
    ivj_ctx() { g_live_contexts.fetch_add(1, std::memory_order_relaxed); }
    ~ivj_ctx() { g_live_contexts.fetch_sub(1, std::memory_order_relaxed); }
    ivj_ctx(const ivj_ctx&) = delete;
    ivj_ctx& operator=(const ivj_ctx&) = delete;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    Arena arena;
    // state handed from ivj_overlap_count_dev to ivj_overlap_fill_dev
    char* ov_buf = nullptr;
    size_t ov_cap = 0;
    int64_t ov_n = -1;
    const void* ov_probe_start = nullptr;
    const void *ov_probe_contig = nullptr, *ov_probe_end = nullptr;
    const ivj_index* ov_ix = nullptr;
    int32_t ov_filter = -1;
    int32_t* ov_hi = nullptr;
    int32_t* ov_cnt = nullptr;
    long long* ov_tile = nullptr;   // ntiles + 1: tile bases, last = total
    long long* h_total = nullptr;   // pinned
    // Host words (round 5): 64 pinned, host-coherent 32-bit words that KERNELS write with system-scope stores -- the few values the host
    // needs mid-call (largest bucket of the balanced index build, far-row count of the slice tables, pairs + flags of the fused join)
    // arrive without a copy operation in the stream; the host reads them after the event / synchronisation it waited on anyway and
    // falls back to a copy when a word's sequence number is not the call's (IVJ_HOST_WORDS=0: always the copies).
    //   [0] ix3 bad  [1] ix3 largest bucket | merge shift << 24  [2] seq     [4] far rows  [5] seq     [8..9] pairs  [10..11] flags  [12] seq
    uint32_t* hw = nullptr;         // host address
    uint32_t* hw_dev = nullptr;     // the same words as the device sees them
    uint32_t hw_seq = 0;
    uint32_t cs_far_hw_seq = 0;     // sequence number the pending far-row count was written under (0: it travels by copy)
    uint32_t cs_fused_hw_seq = 0;   // sequence number of the running fused join's state words (0: by copy)
    int env_spin_us = 4000;         // IVJ_SPIN_US: the host polls a stream / event this long (microseconds) before it blocks in the runtime's wait; 0: block at once
    long long hw_misses = 0;        // host words that did not carry the call's sequence number (the value was copied instead)
    bool cs_prep_zero = false;      // this call's k_cs_prep clears the call state and the sample histogram (no memsets queued)
    XferSlots xfer;                 // pinned staging slots of the host <-> HBM copies (HostXfer), allocated on first use
    // one released index slab kept for reuse (bench/streaming loops rebuild the index every call)
    char* nl_cache = nullptr;          // the nearest-line table of the last index freed on this context (reused like ix_cache)
    size_t nl_cache_cap = 0;
    int env_nearest_lines = -1;        // IVJ_NEAREST_LINES: 0 never, 1 wherever the kernel applies, unset: by size
    char* ix_cache = nullptr;
    size_t ix_cache_cap = 0;
    int64_t ov_total = 0;
    // bucketed (partitioned) copies of the probe columns + their row ids, when the partition path ran
    bool ov_part = false;
    int32_t *pt_c = nullptr, *pt_s = nullptr, *pt_e = nullptr, *pt_row = nullptr;
    uint32_t* pt_off = nullptr; int pt_ntiles = 0;   // bucket-major scanned tile histogram of the last one-level partition
    uint32_t* pt_bstart = nullptr;     // PART_BUCKETS + 1 bucket starts of the last one-level partition
    bool part_attr_set = false;
    bool os_attr_set = false;
    hipEvent_t cs_event = nullptr;     // marks the read-back of an index's far-row count (host_cslice.hip.h: cs_ensure_tables / cs_resolve_tables)
    const ivj_index* cs_far_owner = nullptr;   //   ... whose count the pinned slot holds
    hipEvent_t ix3_event = nullptr;    // marks the read-back of the balanced build's {bad, largest bucket}
    bool ix3_attr_set = false;         // index build, round 5 (ixsort3.hip.h): LDS attributes set once
    int env_ix_merge = -1;             // IVJ_IX_MERGE: upper bound of the balanced build's merge shift (0 = always 2048 buckets); -1: up to V3_MAX_MERGE
    int env_ix_stage = -1;             // IVJ_IX_STAGE: the balanced build's local kernel with (1) / without (0) the rows staged in LDS; -1 by bucket size
    int env_ix_v3 = -1;                // IVJ_IX_V3: -1 by size, 0 never (the round-2 LSD sort), 1 wherever it applies
    int ix3_fallback_streak = 0;       // consecutive balanced builds handed back; from 2 on the balanced build is skipped except every 16th time
    unsigned ix3_builds_since_fallback = 0;
    int64_t ix3_fallbacks = 0;         // builds the balanced pass handed back to the LSD sort (a bucket above V3_CAP rows, keys beyond 32 bits)
    // slice path (host_slice.hip.h): bucket-ordered probe records, histogram, chunk table, tile totals
    // staging of the last closed streaming session, kept for the next one (pinned allocations cost ~70 ms per GB)
    struct StreamBufs { int32_t* h_in = nullptr; int32_t* d_in = nullptr; size_t in_cap = 0; char* d_out = nullptr; size_t d_out_cap = 0;
                        char* h_out = nullptr; size_t h_out_cap = 0; } st_cache[3];
    char* lb_buf = nullptr; size_t lb_cap = 0;        // status words + ticket of the single-launch look-back scans (lb_scan_u32)
    char* sl_buf = nullptr;
    size_t sl_cap = 0;
    int4* sl_rec = nullptr;
    uint2* sl_cache = nullptr;
    uint32_t *sl_gh = nullptr, *sl_rstart = nullptr, *sl_rcur = nullptr, *sl_bend = nullptr;   // sampled partition (cslice.hip.h)
    bool sl_sampled = false;           // the last contig-aligned partition was the sampled one (the join reads sl_bend)
    uint32_t *sl_blk = nullptr, *sl_part = nullptr, *sl_bstart = nullptr;
    int32_t* sl_meta = nullptr;
    int2* sl_map = nullptr;
    long long *sl_tile = nullptr, *sl_tpart = nullptr;
    bool ov_slice = false;             // the pending count -> fill hand-over went through the slice path
    bool ov_cs = false;                //   ... through its contig-aligned form (cslice.hip.h)
    bool sl_plan_valid = false;
    int sl_items = 2;                  // probes per thread of the slice join (IVJ_SLICE_ITEMS = 2 | 4: tuning knob)
    int env_joint_bins = 0, env_count_nolds = 0, env_count_ablate = 0;     // IVJ_JOINT_BINS (1|2), IVJ_COUNT_NOLDS: tuning knobs of count_overlaps
    int sl_env_rows = 0, sl_env_chunk = 0, sl_env_notab = 0, sl_env_nobins = 0, sl_env_ablate = 0, sl_env_auto = 1, sl_env_stable = 0, sl_env_sthreads = 1024;   // IVJ_SLICE_ROWS / IVJ_SLICE_CHUNK: tuning knobs used when the opts fields are 0
    SlicePlan sl_plan;
    bool cs_sattr_set = false;         // ... and those of the sampled partition
    bool cs_attr_set = false;          // contig-aligned slice path (cslice.hip.h): LDS attributes set once
    int cs_env_walk = -1;           // IVJ_CS_WALK: -1 auto, 0 / 1 force the join kernel of the contig-aligned slices
    int cs_env_off = 0;                // IVJ_CS=0: keep the round-2 slice kernels (A/B runs)
    int cs_env_ptile = 0;              // IVJ_CS_PTILE=4096: partition tiles of 4096 probes even where 8192 fit
    int cs_env_fill_two = 0;           // IVJ_CS_FILL_TWO=1: k_cs_fill with two workgroups per CU and no prefetch (A/B runs: 0.68-0.72 against 0.63-0.69 ms)
    int cs_env_sampled = 1;            // IVJ_CS_SAMPLED=0: always the histogram-first partition (A/B runs)
    int cs_env_slack = 0;              // IVJ_CS_SLACK: records of slack per bucket region (tests force the overflow path with a tiny one)
    int64_t cs_sampled_overflows = 0;  // calls redone because a sampled region overflowed (ivj_debug_counter)
    bool cs_force_exact = false;       // set while a call whose sampled regions overflowed is redone
    int cs_env_fuse_sample = 1;        // IVJ_CS_FUSE_SAMPLE=0: the slice bins and the probe sample of a fresh index as two launches (A/B runs)
    int cs_env_rec8 = 1;               // IVJ_CS_REC8=0: always 12-byte probe records (A/B runs)
    bool cs_force_rec12 = false;       // set while a call whose 8-byte records overflowed is redone
    int cs_rec8_streak = 0;            // consecutive calls whose 8-byte records overflowed (cleared by a call that kept them)
    unsigned cs_rec8_skipped = 0;      // calls that did not try the 8-byte form because of the streak
    bool cs_rec8_tried = false;        // the call in flight offered the 8-byte form
    int64_t cs_rec8_overflows = 0;     // calls redone with 12-byte records
    int cs_env_nocache = 0;            // IVJ_CS_NOCACHE=1: the FILL pass matches again instead of reading COUNT's words (A/B runs)
    int cs_env_persist = 1;            // IVJ_CS_PERSIST=0: one join workgroup per list item instead of persistent ones (rounds 3-5; A/B runs)
    int cs_env_pmax = 0, cs_env_pgrain = 0;   // IVJ_CS_PMAX / IVJ_CS_PGRAIN: items per draw of a persistent workgroup (defaults 4 / 64)
    int cs_env_pchunks = 0;            // IVJ_CS_PCHUNKS: scatter workgroups of the wide-tile sampled partition (0: ~ 2048)
    int n_cus = 0;                     // compute units of the device
    const char* cs_env_wgtrace = nullptr;   // IVJ_CS_WGTRACE=<file>: the fused plain join records its workgroups' time line (diagnosis; tools/wgtrace.py)
    unsigned long long* cs_trace_buf = nullptr;
    size_t cs_trace_cap = 0;           // workgroups the buffer holds
    // timing
    int timing = 0;          // 0 off, 1 probe kernels only, 2 every kernel
    bool t_open = false;
    std::vector<TimingRec> recs;
    std::vector<hipEvent_t> pool;
    size_t pool_used = 0;
    // indexes built on this context that are still alive: ivj_ctx_destroy detaches them (ix->ctx = nullptr), so an
    // ivj_index_free that comes after the context is gone only releases the index's own slab
    std::vector<ivj_index*> live;
    // streaming sessions opened on this context that are still alive: ivj_ctx_destroy releases and detaches them
    std::vector<struct ivj_stream*> streams;
    // communicators created on this context that are still alive: ivj_ctx_destroy detaches them (their handles stay destroyable)
    std::vector<struct ivj_comm*> comms;
};

// Waits of the hot path (round 5).  The runtime's hipStreamSynchronize / hipEventSynchronize sleep on the completion interrupt: tens of
// microseconds between the kernel's end and the host's next launch -- per call, with the GPU idle.  The host polls the stream / event
// for up to IVJ_SPIN_US microseconds first (default 4000: a call of the benchmark configurations ends within it
// -- a call still running after that blocks in the runtime as before).  Same completion semantics as the runtime's wait.
// Round 6: the poll loop executes `pause` between queries, and in a process that holds SEVERAL contexts (MultiEngine's per-device host
// threads, the 8-rank dry run) it yields the core every few polls and polls for at most IVJ_SPIN_US / 8: N spinning waiters under a
// cgroup CPU quota otherwise starve the producer threads (host_copy_parallel, Arrow assembly) they are waiting for.
inline void spin_relax(int i, bool crowded) {
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(__i386__))
    __builtin_ia32_pause();
#endif
    if (crowded && (i & 7) == 7) std::this_thread::yield();
}
template <typename Query>
inline bool spin_wait(ivj_ctx* ctx, Query&& q, hipError_t* out) {
    if (ctx->env_spin_us <= 0) return false;
    const bool crowded = g_live_contexts.load(std::memory_order_relaxed) > 1;
    const long long budget = crowded ? std::max(ctx->env_spin_us / 8, 1) : ctx->env_spin_us;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0;; ++i) {
        const hipError_t e = q();
        if (e != hipErrorNotReady) { *out = e; return true; }
        if ((i & 15) == 15 && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > budget) break;
        spin_relax(i, crowded);
    }
    (void)hipGetLastError();                                                   // (hipErrorNotReady is sticky in hipGetLastError)
    return false;
}
inline hipError_t wait_stream(ivj_ctx* ctx, hipStream_t s) {
    hipError_t e;
    if (spin_wait(ctx, [&] { return hipStreamQuery(s); }, &e)) return e;
    return hipStreamSynchronize(s);
}
inline hipError_t wait_event(ivj_ctx* ctx, hipEvent_t ev) {
    hipError_t e;
    if (spin_wait(ctx, [&] { return hipEventQuery(ev); }, &e)) return e;
    return hipEventSynchronize(ev);
}

struct ivj_index {
    ivj_ctx* ctx = nullptr;
    int device = 0;
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
    char* ix3_z = nullptr;           // zeroed words of the balanced index build inside the slab's zeroed head (nullptr: the build takes them from the arena)
    int4* nrec = nullptr;
    int4* cmeta_j = nullptr;
    int4* crec = nullptr;
    int64_t bins_len = 0;
    bool has_end_order = false, has_end_table = false;
    bool has_argmax = false;
    int4* nline = nullptr;             // nearest lines (build_lines): own allocation, 128 bytes per table slot, on first use
    size_t nline_cap = 0;
    bool has_lines = false;
    bool has_flat = false;
    bool has_rec4 = false;
    unsigned long long* spl = nullptr;   // slice path: composite key of the first row of every slice
    int sl_R = 0, sl_nb = 0, sl_ncells = -1;   //   geometry the splitters were made for (0: none yet)
    int4* sl_cm = nullptr;               //   direct-address table over the splitters: per-contig grid, cells
    uint32_t* sl_cell = nullptr;
    // contig-aligned slice path (cslice.hip.h): geometry fixed at build time, arrays in the slab, filled on first use
    CsGeom cs_g{0, 0, 0, 0, 0};
    bool cs_ok = false, cs_built = false;
    bool cs_far_pending = false;         // the far-row count is on its way to the host: cs_walk is decided by cs_resolve_tables
    bool cs_walk = false;                // k_cs_join (windows that run on walk the block maxima) instead of k_cs_join_plain
    int64_t cs_far = 0;                  // rows whose window would overflow the branch-free one (k_cs_bins)
    int32_t* cs_bound = nullptr;
    unsigned long long* cs_spl = nullptr;
    int4* cs_cm = nullptr;
    uint32_t* cs_cell = nullptr;
    unsigned short* cs_bins = nullptr;
    int4* cs_smeta = nullptr;
    int32_t* hier = nullptr;             // the sorted ends and their block maxima, level by level (k_hier_level); filled on first use
    bool hier_built = false;
    bool tables_built = false;   // the direct-address tables exist (built on first use)
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
           !std::strncmp(name, "subtract_", 9) || !std::strncmp(name, "cluster_", 8) || !std::strncmp(name, "part_scatter", 12) ||
           !std::strncmp(name, "slice_", 6) || !std::strncmp(name, "cs_", 3);
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

// Single-launch (decoupled look-back) scan of a u32 array in place: one memset of the status words + one kernel instead of the
// three launches of device_scan.  The status buffer is owned by the context and reused by successive scans (stream order).
template <class Op, bool EXCLUSIVE>
int lb_scan_u32(ivj_ctx* ctx, const char* name, uint32_t* data, int64_t n, uint32_t identity) {
    if (n <= 0) return IVJ_OK;
    const int64_t tiles = (n + LB_TILE - 1) / LB_TILE;
    const size_t need = align_up((size_t)tiles * 8) + align_up(16);
    if (need > ctx->lb_cap) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->lb_buf) HIP_TRY(hipFree(ctx->lb_buf));
        ctx->lb_buf = nullptr; ctx->lb_cap = 0;
        const size_t want = align_up(need + need / 2, 1 << 16);
        hipError_t e = hipMalloc((void**)&ctx->lb_buf, want);
        if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string("look-back scan status hipMalloc: ") + hipGetErrorString(e));
        ctx->lb_cap = want;
    }
    HIP_TRY(hipMemsetAsync(ctx->lb_buf, 0, need, ctx->stream));
    LAUNCH(ctx, name, (k_scan_lb_u32<Op, EXCLUSIVE>), tiles, OS_THREADS, data, n, identity,
           (uint32_t*)(ctx->lb_buf + align_up((size_t)tiles * 8)), (unsigned long long*)ctx->lb_buf);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
}

// look-back status words + ticket of one single-launch scan (zeroed); nullptr when they cannot be had
char* lb_status(ivj_ctx* ctx, int64_t tiles) {
    const size_t need = align_up((size_t)tiles * 8) + align_up(16);
    if (need > ctx->lb_cap) {
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return nullptr;
        if (ctx->lb_buf) (void)hipFree(ctx->lb_buf);
        ctx->lb_buf = nullptr; ctx->lb_cap = 0;
        const size_t want = align_up(need + need / 2, 1 << 16);
        if (hipMalloc((void**)&ctx->lb_buf, want) != hipSuccess) { ctx->lb_buf = nullptr; return nullptr; }
        ctx->lb_cap = want;
    }
    if (hipMemsetAsync(ctx->lb_buf, 0, need, ctx->stream) != hipSuccess) return nullptr;
    return ctx->lb_buf;
}

// device-wide scan.  Sums of uint32 / int64 (every use but the max scans of the tables): ONE launch, decoupled look-back (round 4);
// anything else, or no status buffer: three launches (reduce, partials, apply).
template <class T, class Op, bool INCLUSIVE>
void device_scan(ivj_ctx* ctx, const char* name, const T* in, T* out, int64_t n, T identity, T* partials, T* total_out) {
    if constexpr (std::is_same<Op, SumOp>::value && (std::is_same<T, uint32_t>::value || std::is_same<T, long long>::value)) {
        if (n > 0 && identity == (T)0) {
            const bool big = n >= (4ll << 20);
            const int64_t tile = (int64_t)OS_THREADS * (big ? 32 : LB_ITEMS), lt = (n + tile - 1) / tile;
            if (char* st = lb_status(ctx, lt)) {
                if (big) LAUNCH(ctx, name, (k_scan_lb_sum<T, INCLUSIVE, 32>), lt, OS_THREADS, in, out, n, (uint32_t*)(st + align_up((size_t)lt * 8)), (unsigned long long*)st, total_out);
                else LAUNCH(ctx, name, (k_scan_lb_sum<T, INCLUSIVE, LB_ITEMS>), lt, OS_THREADS, in, out, n, (uint32_t*)(st + align_up((size_t)lt * 8)), (unsigned long long*)st, total_out);
                return;
            }
        }
    }
    const int64_t tiles = scan_num_tiles(n);
    LAUNCH(ctx, name, (k_scan_reduce<T, Op>), tiles, SCAN_THREADS, in, n, identity, partials);
    LAUNCH(ctx, name, (k_scan_partials<T, Op>), 1, SCAN_THREADS, partials, tiles, identity, total_out);
    LAUNCH(ctx, name, (k_scan_apply<T, Op, INCLUSIVE>), tiles, SCAN_THREADS, in, out, n, identity, (const T*)partials);
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
    if (o->table_mode < 0 || o->table_mode > 3) return fail(IVJ_EINVAL, "table_mode must be 0 (auto), 1 (records), 2 (bins) or 3 (records + nearest lines)");
    if (o->partition_mode < 0 || o->partition_mode > 6 || o->partition_mode == 3 || o->partition_mode == 4)
        return fail(IVJ_EINVAL, "partition_mode must be 0 (auto), 1 (256-way buckets), 2 (never), 5 (flat, fused path only) or 6 (LDS-resident index slices)");
    if (o->slice_rows < 0 || o->slice_chunk < 0) return fail(IVJ_EINVAL, "slice_rows / slice_chunk must be >= 0");
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
    v.cmeta = ix->cmeta; v.brec = ix->brec; v.cmeta_e = ix->cmeta_e; v.brec_e = ix->brec_e; v.pargmax = ix->pargmax; v.nrec = ix->nrec; v.nline = ix->nline; v.cmeta_j = ix->cmeta_j; v.crec = ix->crec;
    v.bins = ix->bins; v.bins_e = ix->bins_e; v.rec4 = ix->rec4; v.tab2 = ix->tab2;
    // 16-byte bin records once the 4-byte tables + key arrays no longer fit the XCD L2s anyway
    {
        const HierShape h = hier_shape(ix->n > 0 ? ix->n : 1);
        v.hier.v = ix->hier; v.hier.nlev = h.nlev;
        for (int l = 0; l < HIER_MAX; ++l) v.hier.off[l] = h.off[l];
    }
    v.use_rec = (ix->table_mode == 1 || ix->table_mode == 3) ? 1 : (ix->table_mode == 2 ? 0 : (ix->n >= (1ll << 20) ? 1 : 0));
    return v;
}


}  // namespace
