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

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <thread>
#include <type_traits>

#include "probe.hip.h"
#include "slice.hip.h"
#include "cslice.hip.h"
#include "onesweep.hip.h"
#include "ixsort3.hip.h"
#include "scan.hip.h"

using namespace ivj;

#include "host_core.hip.h"
#include "host_mem.hip.h"
#include "host_index.hip.h"
#include "host_slice.hip.h"
#include "host_cslice.hip.h"
#include "host_join.hip.h"
#include "host_sortscan.hip.h"
#include "host_stream.hip.h"
#include "host_comm.hip.h"

// =============================================================================== C ABI

extern "C" {

const char* ivj_last_error(void) { return g_err.c_str(); }
const char* ivj_version(void) { return "ivjoin-hip 0.1 (gfx950)"; }
int ivj_abi_version(void) { return IVJ_ABI_VERSION; }
int64_t ivj_host_mem_available(void) { return (int64_t)host_mem_available(); }

int ivj_device_count(int* n) try {
    if (!n) return fail(IVJ_EINVAL, "n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail(IVJ_EHIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *n = c;
    return IVJ_OK;
} IVJ_ABI_CATCH

int ivj_ctx_create(int device, ivj_ctx** out) try {
    if (!out) return fail(IVJ_EINVAL, "out is NULL");
    int cnt = 0;
    IVJ_TRY(ivj_device_count(&cnt));
    if (device < 0 || device >= cnt) return fail(IVJ_EHIP, "no usable HIP device " + std::to_string(device) + " (device count " + std::to_string(cnt) + ")");
    HIP_TRY(hipSetDevice(device));
    ivj_ctx* ctx = new ivj_ctx();
    ctx->device = device;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) ctx->n_cus = cus; }
    hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return fail(IVJ_EHIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    ctx->stream = ctx->own_stream;
    if (const char* ev = std::getenv("IVJ_SLICE_ITEMS")) ctx->sl_items = std::atoi(ev) == 4 ? 4 : 2;
    if (const char* ev = std::getenv("IVJ_SLICE_ROWS")) ctx->sl_env_rows = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_SLICE_CHUNK")) ctx->sl_env_chunk = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_SLICE_NOTAB")) ctx->sl_env_notab = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_SLICE_NOBINS")) ctx->sl_env_nobins = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_SLICE_ABLATE")) ctx->sl_env_ablate = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_SLICE_AUTO")) ctx->sl_env_auto = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_SLICE_STABLE")) ctx->sl_env_stable = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_SLICE_SCATTER_THREADS")) ctx->sl_env_sthreads = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_CS")) ctx->cs_env_off = std::atoi(ev) == 0 ? 1 : 0;
    if (const char* ev = std::getenv("IVJ_CS_PTILE")) ctx->cs_env_ptile = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_CS_NOCACHE")) ctx->cs_env_nocache = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_CS_PERSIST")) ctx->cs_env_persist = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_CS_PCHUNKS")) ctx->cs_env_pchunks = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_CS_PMAX")) ctx->cs_env_pmax = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_CS_PGRAIN")) ctx->cs_env_pgrain = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_CS_WGTRACE")) ctx->cs_env_wgtrace = ev[0] ? ev : nullptr;
    if (const char* ev = std::getenv("IVJ_CS_SAMPLED")) ctx->cs_env_sampled = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_CS_SLACK")) ctx->cs_env_slack = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_CS_REC8")) ctx->cs_env_rec8 = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_CS_FUSE_SAMPLE")) ctx->cs_env_fuse_sample = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_CS_FILL_TWO")) ctx->cs_env_fill_two = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_CS_WALK")) ctx->cs_env_walk = std::atoi(ev) != 0 ? 1 : 0;
    if (const char* ev = std::getenv("IVJ_JOINT_BINS")) ctx->env_joint_bins = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_COUNT_ABLATE")) ctx->env_count_ablate = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_NEAREST_LINES")) ctx->env_nearest_lines = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_COUNT_NOLDS")) ctx->env_count_nolds = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_IX_V3")) ctx->env_ix_v3 = std::atoi(ev) != 0 ? 1 : 0;
    if (const char* ev = std::getenv("IVJ_SPIN_US")) ctx->env_spin_us = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_IX_MERGE")) ctx->env_ix_merge = std::atoi(ev);
    if (const char* ev = std::getenv("IVJ_IX_STAGE")) ctx->env_ix_stage = std::atoi(ev) != 0 ? 1 : 0;
    e = hipHostMalloc((void**)&ctx->h_total, 64, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipStreamDestroy(ctx->own_stream); delete ctx; return fail(IVJ_EHIP, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
    {
        // host words (host_core.hip.h): optional -- without them every value travels by copy as before
        const char* ev = std::getenv("IVJ_HOST_WORDS");
        if (!(ev && std::atoi(ev) == 0)) {
            void* hp = nullptr;
            void* dp = nullptr;
            if (hipHostMalloc(&hp, 256, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
                if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess && dp) {
                    std::memset(hp, 0, 256);
                    ctx->hw = (uint32_t*)hp; ctx->hw_dev = (uint32_t*)dp;
                } else (void)hipHostFree(hp);
            }
            (void)hipGetLastError();
        }
    }
    *out = ctx;
    return IVJ_OK;
} IVJ_ABI_CATCH

namespace {
void stream_release(ivj_stream* st, bool keep_cache);
void free_stream_bufs(ivj_ctx::StreamBufs& b) {
    if (b.h_in) (void)hipHostFree(b.h_in);
    if (b.d_in) (void)hipFree(b.d_in);
    if (b.d_out) (void)hipFree(b.d_out);
    if (b.h_out) (void)hipHostFree(b.h_out);
    b = ivj_ctx::StreamBufs();
}
}  // namespace

void ivj_ctx_destroy(ivj_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    while (!ctx->streams.empty()) stream_release(ctx->streams.back(), false);   // live streaming sessions: released and detached (their handles stay closable)
    for (ivj_index* ix : ctx->live) ix->ctx = nullptr;          // detached: they keep (and later free) their slabs
    for (ivj_comm* cm : ctx->comms) comm_detach(cm);            // detached: no call runs on them any more, ivj_comm_destroy still releases them
    if (ctx->arena.base) (void)hipFree(ctx->arena.base);
    if (ctx->ov_buf) (void)hipFree(ctx->ov_buf);
    if (ctx->sl_buf) (void)hipFree(ctx->sl_buf);
    if (ctx->cs_trace_buf) (void)hipFree(ctx->cs_trace_buf);
    if (ctx->lb_buf) (void)hipFree(ctx->lb_buf);
    for (auto& cb : ctx->st_cache) free_stream_bufs(cb);
    if (ctx->ix_cache) (void)hipFree(ctx->ix_cache);
    if (ctx->nl_cache) (void)hipFree(ctx->nl_cache);
    if (ctx->ix3_event) (void)hipEventDestroy(ctx->ix3_event);
    if (ctx->cs_event) (void)hipEventDestroy(ctx->cs_event);
    if (ctx->h_total) (void)hipHostFree(ctx->h_total);
    if (std::getenv("IVJ_DEBUG_REDO")) std::fprintf(stderr, "[ivj] context: host words %s, %lld misses of %u; ix3 fallbacks %lld\n", ctx->hw ? "on" : "off", (long long)ctx->hw_misses, ctx->hw_seq, (long long)ctx->ix3_fallbacks);
    if (ctx->hw) (void)hipHostFree(ctx->hw);
    ctx->xfer.release();
    for (hipEvent_t ev : ctx->pool) (void)hipEventDestroy(ev);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int ivj_ctx_set_stream(ivj_ctx* ctx, void* hip_stream) try {
    if (!ctx) return fail(IVJ_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stream = (hip_stream == (void*)-1) ? ctx->own_stream : (hipStream_t)hip_stream;
    return IVJ_OK;
} IVJ_ABI_CATCH

int ivj_ctx_sync(ivj_ctx* ctx) try {
    if (!ctx) return fail(IVJ_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IVJ_OK;
} IVJ_ABI_CATCH

namespace ivj { __global__ void k_profile_mark() {} }

int ivj_ctx_profile_mark(ivj_ctx* ctx) try {
    if (!ctx) return fail(IVJ_EINVAL, "ctx is NULL");
    DeviceGuard g(ctx->device);
    hipLaunchKernelGGL(ivj::k_profile_mark, dim3(1), dim3(1), 0, ctx->stream);
    HIP_TRY(hipGetLastError());
    return IVJ_OK;
} IVJ_ABI_CATCH

int ivj_ctx_enable_timing(ivj_ctx* ctx, int on) try {
    if (!ctx) return fail(IVJ_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->timing = on < 0 ? 0 : (on > 2 ? 2 : on);
    ctx->recs.clear();
    ctx->pool_used = 0;
    return IVJ_OK;
} IVJ_ABI_CATCH

int ivj_ctx_get_timings(ivj_ctx* ctx, ivj_timing* out, int cap, int* n) try {
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
} IVJ_ABI_CATCH

// ---------------------------------------------------------------- device-resident API

int ivj_index_build_dev(ivj_ctx* ctx, const ivj_side* build_dev, const ivj_opts* opts, int with_end_order, ivj_index** out) try {
    if (!ctx || !out) return fail(IVJ_EINVAL, "ctx or out is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(build_dev, "build"));
    DeviceGuard g(ctx->device);
    return index_build(ctx, build_dev, opts, with_end_order, out);
} IVJ_ABI_CATCH

void ivj_index_free(ivj_index* ix) {
    if (!ix) return;
    ivj_ctx* ctx = ix->ctx;
    if (ctx) {
        if (ctx->ov_ix == ix) { ctx->ov_ix = nullptr; ctx->ov_n = -1; }
        for (size_t k = 0; k < ctx->live.size(); ++k)
            if (ctx->live[k] == ix) { ctx->live[k] = ctx->live.back(); ctx->live.pop_back(); break; }
    }
    if (ix->nline) {
        if (ctx && ix->nline_cap > ctx->nl_cache_cap) {
            char* old = ctx->nl_cache;
            ctx->nl_cache = reinterpret_cast<char*>(ix->nline); ctx->nl_cache_cap = ix->nline_cap;
            if (old) { DeviceGuard g(ctx->device); (void)hipFree(old); }
        } else {
            DeviceGuard g(ix->device);
            (void)hipFree(ix->nline);
        }
    }
    if (ix->slab) {
        if (ctx && ix->slab_cap > ctx->ix_cache_cap) {
            // keep the larger slab for the next index on this context (same stream => ordered reuse)
            char* old = ctx->ix_cache;
            ctx->ix_cache = ix->slab; ctx->ix_cache_cap = ix->slab_cap;
            if (old) { DeviceGuard g(ctx->device); (void)hipFree(old); }
        } else {
            DeviceGuard g(ix->device);
            (void)hipFree(ix->slab);
        }
    }
    delete ix;
}

int ivj_overlap_count_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, int64_t* n_pairs) try {
    if (!ctx || !ix || !n_pairs) return fail(IVJ_EINVAL, "ctx, index or n_pairs is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    DeviceGuard g(ctx->device);
    return overlap_count(ctx, ix, probe_dev, opts, n_pairs);
} IVJ_ABI_CATCH

int ivj_overlap_fill_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts,
                         int32_t* probe_idx_dev, int32_t* build_idx_dev, int64_t capacity) try {
    if (!ctx || !ix) return fail(IVJ_EINVAL, "ctx or index is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    DeviceGuard g(ctx->device);
    return overlap_fill(ctx, ix, probe_dev, opts, probe_idx_dev, build_idx_dev, capacity);
} IVJ_ABI_CATCH

int ivj_overlap_fused_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts,
                          int32_t* probe_idx_dev, int32_t* build_idx_dev, int64_t capacity, int64_t* n_pairs) try {
    if (!ctx || !ix || !n_pairs) return fail(IVJ_EINVAL, "ctx, index or n_pairs is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    if (capacity < 0 || (capacity > 0 && (!probe_idx_dev || !build_idx_dev))) return fail(IVJ_EINVAL, "bad output buffers");
    DeviceGuard g(ctx->device);
    return overlap_fused(ctx, ix, probe_dev, opts, probe_idx_dev, build_idx_dev, capacity, n_pairs);
} IVJ_ABI_CATCH

// ---- multi-GPU: communicator, all-gatherv, sharded overlap with the exchange overlapping the join (host_comm.hip.h) ----

int ivj_comm_unique_id(void* id_out) try {
    if (!id_out) return fail(IVJ_EINVAL, "id_out is NULL");
    const RcclApi* api = rccl_api();
    if (!api) return fail(IVJ_EHIP, g_rccl.error);
    RcclUniqueId id;
    RCCL_TRY(api, api->GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return IVJ_OK;
} IVJ_ABI_CATCH

int ivj_comm_create(ivj_ctx* ctx, const void* unique_id, int rank, int world, ivj_comm** out) try {
    if (!ctx || !out) return fail(IVJ_EINVAL, "ctx or out is NULL");
    if (world < 1 || rank < 0 || rank >= world) return fail(IVJ_EINVAL, "rank / world out of range");
    if (world > 1 && !unique_id) return fail(IVJ_EINVAL, "unique_id is NULL");
    DeviceGuard g(ctx->device);
    ivj_comm* c = new ivj_comm();
    c->ctx = ctx; c->rank = rank; c->world = world;
    // IVJ_COMM_NO_SHORTCUT=1: also a single rank gets a real RCCL communicator and exchanges with itself through it (tests:
    // the dlopen'ed entry points, the exchange stream's ordering and the error mapping run on a 1-GPU box)
    const char* ns = std::getenv("IVJ_COMM_NO_SHORTCUT");
    c->self_rccl = ns && std::atoi(ns) != 0;
    if (world > 1 || c->self_rccl) {
        const RcclApi* api = rccl_api();
        if (!api) { delete c; return fail(IVJ_EHIP, g_rccl.error); }
        RcclUniqueId id;
        if (unique_id) std::memcpy(&id, unique_id, sizeof(id));
        else {
            const int ur = api->GetUniqueId(&id);
            if (ur != 0) { delete c; return fail(IVJ_EHIP, std::string("ncclGetUniqueId: ") + api->GetErrorString(ur)); }
        }
        const int r = api->CommInitRank(&c->comm, world, id, rank);
        if (r != 0) { delete c; return fail(IVJ_EHIP, std::string("ncclCommInitRank: ") + api->GetErrorString(r)); }
    }
    const int rc = comm_finish_create(c);
    if (rc != IVJ_OK) { ivj_comm_destroy(c); return rc; }
    *out = c;
    return IVJ_OK;
} IVJ_ABI_CATCH

int ivj_comm_create_local(ivj_ctx* const* ctxs, int n, ivj_comm** out) try {
    if (!ctxs || !out || n < 1) return fail(IVJ_EINVAL, "ctxs / out is NULL or n < 1");
    std::vector<RcclComm> comms((size_t)n, nullptr);
    LoopGroup* loop = nullptr;
    if (n > 1) {
        std::vector<int> devs((size_t)n);
        bool shared = false;
        for (int i = 0; i < n; ++i) {
            if (!ctxs[i]) return fail(IVJ_EINVAL, "a context is NULL");
            devs[i] = ctxs[i]->device;
            for (int j = 0; j < i; ++j) if (devs[j] == devs[i]) shared = true;
        }
        const char* ev = std::getenv("IVJ_COMM_LOOPBACK");
        if (shared || (ev && std::atoi(ev) != 0)) {
            // RCCL needs one device per rank; ranks that share a device (a 1-GPU box running the world-n protocol), or a host
            // that asks for it, get the in-process transport
            loop = new LoopGroup();
            loop->world = n; loop->refs = n;
            loop->vals.assign((size_t)n * 2, 0);
            loop->send.assign((size_t)n * LoopGroup::MAX_COLS, nullptr);
            loop->cnt.assign((size_t)n, 0);
        } else {
            const RcclApi* api = rccl_api();
            if (!api) return fail(IVJ_EHIP, g_rccl.error);
            RCCL_TRY(api, api->CommInitAll(comms.data(), n, devs.data()));
        }
    }
    for (int i = 0; i < n; ++i) {
        ivj_comm* c = new ivj_comm();
        c->ctx = ctxs[i]; c->rank = i; c->world = n; c->comm = comms[i]; c->loop = loop;
        const int rc = comm_finish_create(c);
        if (rc != IVJ_OK) {
            const std::string why = g_err;
            ivj_comm_destroy(c);
            for (int j = 0; j < i; ++j) { ivj_comm_destroy(out[j]); out[j] = nullptr; }
            if (loop && n - 1 - i > 0) {                 // the communicators never made: drop their references too
                bool last;
                { std::lock_guard<std::mutex> lk(loop->mu); loop->refs -= n - 1 - i; last = loop->refs == 0; }
                if (last) delete loop;
            }
            g_err = why;
            return rc;
        }
        out[i] = c;
    }
    return IVJ_OK;
} IVJ_ABI_CATCH

void ivj_comm_destroy(ivj_comm* c) {
    if (!c) return;
    DeviceGuard g(c->device);
    if (c->ctx) { auto& v = c->ctx->comms; v.erase(std::remove(v.begin(), v.end(), c), v.end()); }
    if (c->xstream) { (void)hipStreamSynchronize(c->xstream); }
    if (c->comm && rccl_api()) (void)rccl_api()->CommDestroy(c->comm);
    if (c->loop) {
        bool last;
        { std::lock_guard<std::mutex> lk(c->loop->mu); last = --c->loop->refs == 0; }
        if (last) delete c->loop;
    }
    for (auto& b : c->stage) if (b) (void)hipFree(b);
    if (c->iota) (void)hipFree(c->iota);
    if (c->pp_buf) (void)hipFree(c->pp_buf);
    if (c->d_counts) (void)hipFree(c->d_counts);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    delete c;
}

int ivj_comm_info(const ivj_comm* c, int* rank, int* world) try {
    if (!c) return fail(IVJ_EINVAL, "comm is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return IVJ_OK;
} IVJ_ABI_CATCH

int ivj_allgather_counts(ivj_comm* c, int64_t n_local, int64_t* counts) try {
    if (!c || !counts || n_local < 0) return fail(IVJ_EINVAL, "comm / counts is NULL or n_local < 0");
    if (!c->ctx) return fail(IVJ_ESTATE, "the communicator's context was destroyed");
    DeviceGuard g(c->ctx->device);
    return comm_allgather_counts(c, n_local, counts);
} IVJ_ABI_CATCH

int ivj_allgatherv_dev(ivj_comm* c, const void* const* send_cols, void* const* recv_cols, int n_cols, int elem_bytes, const int64_t* counts) try {
    if (!c || !send_cols || !recv_cols || !counts || n_cols < 1 || elem_bytes < 1) return fail(IVJ_EINVAL, "bad all-gatherv arguments");
    if (!c->ctx) return fail(IVJ_ESTATE, "the communicator's context was destroyed");
    DeviceGuard g(c->ctx->device);
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));                   // the payload is whatever the context's stream produced
    IVJ_TRY(comm_exchange(c, send_cols, recv_cols, n_cols, elem_bytes, counts, 0));
    HIP_TRY(hipStreamSynchronize(c->xstream));
    return IVJ_OK;
} IVJ_ABI_CATCH

int ivj_overlap_allgather_dev(ivj_comm* c, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, int n_chunks,
                              int32_t* probe_idx_dev, int32_t* build_idx_dev, int64_t capacity, int64_t* n_total, int64_t* n_local) try {
    if (!c || !ix || !n_total) return fail(IVJ_EINVAL, "comm, index or n_total is NULL");
    if (!c->ctx) return fail(IVJ_ESTATE, "the communicator's context was destroyed");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    if (n_chunks < 1 || n_chunks > 64) return fail(IVJ_EINVAL, "n_chunks must be in 1 .. 64");
    if (capacity < 0 || (capacity > 0 && (!probe_idx_dev || !build_idx_dev))) return fail(IVJ_EINVAL, "bad output buffers");
    DeviceGuard g(c->ctx->device);
    return overlap_allgather(c, ix, probe_dev, opts, n_chunks, probe_idx_dev, build_idx_dev, capacity, n_total, n_local);
} IVJ_ABI_CATCH

int ivj_count_overlaps_allgather_dev(ivj_comm* c, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, int64_t n_total,
                                     int64_t* counts_dev) try {
    if (!c || !ix) return fail(IVJ_EINVAL, "comm or index is NULL");
    if (!c->ctx) return fail(IVJ_ESTATE, "the communicator's context was destroyed");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    if (n_total < 0 || (n_total > 0 && !counts_dev)) return fail(IVJ_EINVAL, "n_total < 0 or counts is NULL");
    DeviceGuard g(c->ctx->device);
    return per_probe_allgather(c, ix, probe_dev, opts, IVJ_STREAM_COUNT, n_total, counts_dev, nullptr, nullptr, nullptr);
} IVJ_ABI_CATCH

int ivj_nearest_allgather_dev(ivj_comm* c, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, int64_t n_total,
                              int32_t* idx_dev, int64_t* dist_dev, int32_t* n_found_dev) try {
    if (!c || !ix) return fail(IVJ_EINVAL, "comm or index is NULL");
    if (!c->ctx) return fail(IVJ_ESTATE, "the communicator's context was destroyed");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    if (opts->nearest_k > 1024) return fail(IVJ_EINVAL, "nearest_k > 1024");
    if (n_total < 0 || (n_total > 0 && (!idx_dev || !dist_dev || !n_found_dev))) return fail(IVJ_EINVAL, "n_total < 0 or nearest output buffers are NULL");
    DeviceGuard g(c->ctx->device);
    return per_probe_allgather(c, ix, probe_dev, opts, IVJ_STREAM_NEAREST, n_total, nullptr, idx_dev, dist_dev, n_found_dev);
} IVJ_ABI_CATCH

int ivj_overlap_fused_rows_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, const ivj_rows* rows_dev,
                               int64_t* n_pairs) try {
    if (!ctx || !ix || !rows_dev || !n_pairs) return fail(IVJ_EINVAL, "ctx, index, rows or n_pairs is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    if (rows_dev->n_pairs < 0) return fail(IVJ_EINVAL, "capacity (rows->n_pairs) < 0");
    if (opts->partition_mode == 5) return fail(IVJ_EINVAL, "partition_mode 5 is not available for the rows path");
    DeviceGuard g(ctx->device);
    return overlap_fused_rows(ctx, ix, probe_dev, opts, rows_dev, n_pairs);
} IVJ_ABI_CATCH

int ivj_count_overlaps_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, int64_t* counts_dev) try {
    if (!ctx || !ix) return fail(IVJ_EINVAL, "ctx or index is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    if (probe_dev->n > 0 && !counts_dev) return fail(IVJ_EINVAL, "counts is NULL");
    DeviceGuard g(ctx->device);
    return count_overlaps_dev(ctx, ix, probe_dev, opts, counts_dev);
} IVJ_ABI_CATCH

int ivj_nearest_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, int32_t* idx_dev,
                    int64_t* dist_dev, int32_t* n_found_dev) try {
    if (!ctx || !ix) return fail(IVJ_EINVAL, "ctx or index is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    if (probe_dev->n > 0 && (!idx_dev || !dist_dev || !n_found_dev)) return fail(IVJ_EINVAL, "nearest output buffers are NULL");
    if (opts->nearest_k > 1024) return fail(IVJ_EINVAL, "nearest_k > 1024");
    DeviceGuard g(ctx->device);
    return nearest_dev(ctx, ix, probe_dev, opts, idx_dev, dist_dev, n_found_dev);
} IVJ_ABI_CATCH

// ---------------------------------------------------------------- host-buffer API

int ivj_overlap(ivj_ctx* ctx, const ivj_side* probe, const ivj_side* build, const ivj_opts* opts, ivj_pairs* out) try {
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
    if (!host_result_fits((size_t)total * 8))
        return fail(IVJ_ENOMEM, "the result (" + std::to_string(total) + " pairs, " + std::to_string((size_t)total * 8 >> 20) + " MiB) does not fit the available host memory; use the streaming entry points (ivj_stream_*) or the *_dev ones");
    DevBuf op, ob;
    hipError_t e = hipMalloc(&op.p, (size_t)total * 4);
    if (e == hipSuccess) e = hipMalloc(&ob.p, (size_t)total * 4);
    if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string("hipMalloc(pairs): ") + hipGetErrorString(e));
    IVJ_TRY(overlap_fill(ctx, h.ix, &dp.s, opts, (int32_t*)op.p, (int32_t*)ob.p, total));
    out->probe_idx = (int32_t*)host_result_alloc((size_t)total * 4);
    out->build_idx = (int32_t*)host_result_alloc((size_t)total * 4);
    if (!out->probe_idx || !out->build_idx) { ivj_pairs_free(out); return fail(IVJ_ENOMEM, "host malloc(pairs)"); }
    HostXfer copy(ctx->stream, &ctx->xfer);
    copy.d2h(out->probe_idx, op.p, (size_t)total * 4);
    copy.d2h(out->build_idx, ob.p, (size_t)total * 4);
    const hipError_t ce = copy.finish();
    if (ce != hipSuccess) { ivj_pairs_free(out); return fail(IVJ_EHIP, std::string("D2H(pairs): ") + hipGetErrorString(ce)); }
    out->n_pairs = total;
    return IVJ_OK;
} IVJ_ABI_CATCH

// ---------------------------------------------------------------- merge / cluster / coverage

int ivj_cluster_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_opts* opts, int64_t min_dist, int64_t* cluster_dev, int32_t* cluster_start_dev,
                    int32_t* cluster_end_dev, int64_t* n_clusters) try {
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
} IVJ_ABI_CATCH

int ivj_merge_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_opts* opts, int64_t min_dist, int64_t capacity, int32_t* contig_dev, int32_t* start_dev,
                  int32_t* end_dev, int64_t* n_intervals_dev, int64_t* n_merged) try {
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
} IVJ_ABI_CATCH

int ivj_coverage_dev(ivj_ctx* ctx, ivj_index* ix, const ivj_side* probe_dev, const ivj_opts* opts, int64_t* coverage_dev) try {
    if (!ctx || !ix) return fail(IVJ_EINVAL, "ctx or index is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(probe_dev, "probe"));
    if (probe_dev->n > 0 && !coverage_dev) return fail(IVJ_EINVAL, "coverage is NULL");
    DeviceGuard g(ctx->device);
    return coverage_core(ctx, ix, probe_dev, opts, coverage_dev);
} IVJ_ABI_CATCH

void ivj_merged_free(ivj_merged* m) {
    if (!m) return;
    std::free(m->contig); std::free(m->start); std::free(m->end); std::free(m->n_intervals);
    m->contig = m->start = m->end = nullptr; m->n_intervals = nullptr; m->n = 0;
}

int ivj_merge(ivj_ctx* ctx, const ivj_side* side, const ivj_opts* opts, int64_t min_dist, ivj_merged* out) try {
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
    out->contig = (int32_t*)host_result_alloc((size_t)cl.n * 4);
    out->start = (int32_t*)host_result_alloc((size_t)cl.n * 4);
    out->end = (int32_t*)host_result_alloc((size_t)cl.n * 4);
    out->n_intervals = (int64_t*)host_result_alloc((size_t)cl.n * 8);
    if (!out->contig || !out->start || !out->end || !out->n_intervals) { ivj_merged_free(out); return fail(IVJ_ENOMEM, "host malloc(merged)"); }
    HostXfer copy(ctx->stream, &ctx->xfer);
    copy.d2h(out->contig, cl.m_contig, (size_t)cl.n * 4);
    copy.d2h(out->start, cl.m_start, (size_t)cl.n * 4);
    copy.d2h(out->end, cl.m_end, (size_t)cl.n * 4);
    copy.d2h(out->n_intervals, cnt, (size_t)cl.n * 8);
    const hipError_t e = copy.finish();
    if (e != hipSuccess) { ivj_merged_free(out); return fail(IVJ_EHIP, std::string("D2H(merged): ") + hipGetErrorString(e)); }
    out->n = cl.n;
    return IVJ_OK;
} IVJ_ABI_CATCH

int ivj_cluster(ivj_ctx* ctx, const ivj_side* side, const ivj_opts* opts, int64_t min_dist, int64_t* cluster, int32_t* cluster_start,
                int32_t* cluster_end, int64_t* n_clusters) try {
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
    HostXfer copy(ctx->stream, &ctx->xfer);
    copy.d2h(cluster, d_c, n * 8);
    copy.d2h(cluster_start, d_s, n * 4);
    copy.d2h(cluster_end, d_e, n * 4);
    HIP_TRY(copy.finish());
    if (n_clusters) *n_clusters = ncl;
    return IVJ_OK;
} IVJ_ABI_CATCH

int ivj_coverage(ivj_ctx* ctx, const ivj_side* probe, const ivj_side* build, const ivj_opts* opts, int64_t* coverage) try {
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
    HostXfer copy(ctx->stream, &ctx->xfer);
    copy.d2h(coverage, out.p, (size_t)probe->n * 8);
    HIP_TRY(copy.finish());
    return IVJ_OK;
} IVJ_ABI_CATCH

// ---------------------------------------------------------------- subtract / complement

int ivj_subtract_dev(ivj_ctx* ctx, ivj_index* right_ix, const ivj_side* left_dev, const ivj_opts* opts, int64_t capacity, int32_t* row_dev,
                     int32_t* start_dev, int32_t* end_dev, int64_t* n_pieces) try {
    if (!ctx || !right_ix || !n_pieces) return fail(IVJ_EINVAL, "ctx, index or n_pieces is NULL");
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(left_dev, "left"));
    if (capacity < 0) return fail(IVJ_EINVAL, "capacity < 0");
    DeviceGuard g(ctx->device);
    return subtract_core(ctx, right_ix, left_dev, opts, capacity, &row_dev, &start_dev, &end_dev, nullptr, n_pieces);
} IVJ_ABI_CATCH

void ivj_pieces_free(ivj_pieces* p) {
    if (!p) return;
    std::free(p->row); std::free(p->start); std::free(p->end);
    p->row = p->start = p->end = nullptr; p->n = 0;
}

int ivj_subtract(ivj_ctx* ctx, const ivj_side* left, const ivj_side* right, const ivj_opts* opts, ivj_pieces* out) try {
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
    out->row = (int32_t*)host_result_alloc((size_t)total * 4);
    out->start = (int32_t*)host_result_alloc((size_t)total * 4);
    out->end = (int32_t*)host_result_alloc((size_t)total * 4);
    if (!out->row || !out->start || !out->end) { ivj_pieces_free(out); return fail(IVJ_ENOMEM, "host malloc(pieces)"); }
    HostXfer copy(ctx->stream, &ctx->xfer);
    copy.d2h(out->row, d_row, (size_t)total * 4);
    copy.d2h(out->start, d_start, (size_t)total * 4);
    copy.d2h(out->end, d_end, (size_t)total * 4);
    const hipError_t e = copy.finish();
    if (e != hipSuccess) { ivj_pieces_free(out); return fail(IVJ_EHIP, std::string("D2H(pieces): ") + hipGetErrorString(e)); }
    out->n = total;
    return IVJ_OK;
} IVJ_ABI_CATCH

int ivj_complement(ivj_ctx* ctx, const ivj_side* frame, const ivj_side* view, const ivj_opts* opts, ivj_pieces* out) try {
    return ivj_subtract(ctx, view, frame, opts, out);      // the gaps of `frame` inside every view interval
} IVJ_ABI_CATCH

// ---------------------------------------------------------------- row materialisation

int ivj_materialize_dev(ivj_ctx* ctx, const ivj_side* probe_dev, const ivj_side* build_dev, const ivj_rows* rows) try {
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
} IVJ_ABI_CATCH

int ivj_take_dev(ivj_ctx* ctx, const void* src_dev, int32_t elem_bytes, const int32_t* idx_dev, int64_t n, void* dst_dev,
                 uint64_t* validity_dev) try {
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
} IVJ_ABI_CATCH

int ivj_take(ivj_ctx* ctx, const int32_t* idx, int64_t n, int32_t n_cols, const void* const* src, const int64_t* src_rows,
             const int32_t* elem_bytes, void* const* dst, uint64_t* const* validity) try {
    if (!ctx) return fail(IVJ_EINVAL, "ctx is NULL");
    if (n < 0 || n_cols < 0) return fail(IVJ_EINVAL, "take: negative size");
    if (n == 0 || n_cols == 0) return IVJ_OK;
    if (!idx || !src || !src_rows || !elem_bytes || !dst) return fail(IVJ_EINVAL, "take: NULL argument");
    size_t max_src = 0, max_dst = 0;
    bool any_valid = false;
    for (int c = 0; c < n_cols; ++c) {
        if (elem_bytes[c] != 4 && elem_bytes[c] != 8) return fail(IVJ_EINVAL, "take: elem_bytes must be 4 or 8");
        if (src_rows[c] < 0 || !dst[c] || (src_rows[c] > 0 && !src[c])) return fail(IVJ_EINVAL, "take: bad column");
        max_src = std::max(max_src, (size_t)src_rows[c] * (size_t)elem_bytes[c]);
        max_dst = std::max(max_dst, (size_t)n * (size_t)elem_bytes[c]);
        any_valid |= validity && validity[c];
    }
    DeviceGuard g(ctx->device);
    const size_t words = (size_t)((n + 63) / 64);
    DevBuf d_idx, d_src, d_dst, d_val;
    hipError_t e = hipMalloc(&d_idx.p, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc(&d_src.p, max_src ? max_src : 16);
    if (e == hipSuccess) e = hipMalloc(&d_dst.p, max_dst);
    if (e == hipSuccess && any_valid) e = hipMalloc(&d_val.p, words * 8);
    if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string("hipMalloc(take): ") + hipGetErrorString(e));
    HostXfer copy(ctx->stream, &ctx->xfer);
    copy.h2d(d_idx.p, idx, (size_t)n * 4);
    for (int c = 0; c < n_cols; ++c) {
        const size_t sb = (size_t)src_rows[c] * (size_t)elem_bytes[c], db = (size_t)n * (size_t)elem_bytes[c];
        uint64_t* val = validity ? validity[c] : nullptr;
        host_prefault(dst[c], db);
        copy.h2d(d_src.p, src[c], sb);
        HIP_TRY(copy.err);
        int rc = ivj_take_dev(ctx, d_src.p, elem_bytes[c], (const int32_t*)d_idx.p, n, d_dst.p, val ? (uint64_t*)d_val.p : nullptr);
        if (rc != IVJ_OK) return rc;
        copy.d2h(dst[c], d_dst.p, db);
        if (val) copy.d2h(val, d_val.p, words * 8);
        HIP_TRY(copy.finish());                                              // before the device buffers are reused
    }
    HIP_TRY(copy.finish());
    return IVJ_OK;
} IVJ_ABI_CATCH

void ivj_rows_free(ivj_rows* r) {
    if (!r) return;
    int32_t** cols[7] = {&r->probe_idx, &r->build_idx, &r->contig, &r->start_1, &r->end_1, &r->start_2, &r->end_2};
    for (auto c : cols) { std::free(*c); *c = nullptr; }
    r->n_pairs = 0;
}

int ivj_overlap_rows(ivj_ctx* ctx, const ivj_side* probe, const ivj_side* build, const ivj_opts* opts, ivj_rows* out) try {
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
    if (!host_result_fits((size_t)total * 28))
        return fail(IVJ_ENOMEM, "the result (" + std::to_string(total) + " rows x 7 columns) does not fit the available host memory");
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
    HostXfer copy(ctx->stream, &ctx->xfer);
    for (int k = 0; k < 7; ++k) {
        *hcols[k] = (int32_t*)host_result_alloc((size_t)total * 4);
        if (!*hcols[k]) { (void)copy.finish(); ivj_rows_free(out); return fail(IVJ_ENOMEM, "host malloc(rows)"); }
        copy.d2h(*hcols[k], *dcols[k], (size_t)total * 4);
    }
    const hipError_t se = copy.finish();
    if (se != hipSuccess) { ivj_rows_free(out); return fail(IVJ_EHIP, std::string("D2H(rows): ") + hipGetErrorString(se)); }
    out->n_pairs = total;
    return IVJ_OK;
} IVJ_ABI_CATCH

#include "arrow_cdata.hip.h"

void ivj_pairs_free(ivj_pairs* p) {
    if (!p) return;
    std::free(p->probe_idx); std::free(p->build_idx);
    p->probe_idx = nullptr; p->build_idx = nullptr; p->n_pairs = 0;
}

int ivj_count_overlaps(ivj_ctx* ctx, const ivj_side* probe, const ivj_side* build, const ivj_opts* opts, int64_t* counts) try {
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
    HostXfer copy(ctx->stream, &ctx->xfer);
    copy.d2h(counts, dc.p, (size_t)probe->n * 8);
    HIP_TRY(copy.finish());
    return IVJ_OK;
} IVJ_ABI_CATCH

int ivj_nearest(ivj_ctx* ctx, const ivj_side* probe, const ivj_side* build, const ivj_opts* opts, int32_t* idx, int64_t* dist,
                int32_t* n_found) try {
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
    HostXfer copy(ctx->stream, &ctx->xfer);
    copy.d2h(idx, di.p, slots * 4);
    copy.d2h(dist, dd.p, slots * 8);
    copy.d2h(n_found, dn.p, (size_t)probe->n * 4);
    HIP_TRY(copy.finish());
    return IVJ_OK;
} IVJ_ABI_CATCH

// ---------------------------------------------------------------- streaming probe session

int ivj_stream_open(ivj_ctx* ctx, const ivj_side* build, const ivj_opts* opts, int op, int64_t max_batch_rows, ivj_stream** out) try {
    if (!ctx || !out) return fail(IVJ_EINVAL, "ctx or out is NULL");
    *out = nullptr;
    IVJ_TRY(check_opts(opts));
    IVJ_TRY(check_side(build, "build"));
    if (op < IVJ_STREAM_OVERLAP || op > IVJ_STREAM_NEAREST) return fail(IVJ_EINVAL, "op must be IVJ_STREAM_OVERLAP, _COUNT or _NEAREST");
    if (max_batch_rows < 1 || max_batch_rows > 0x7fff0000ll) return fail(IVJ_EINVAL, "max_batch_rows out of range");
    const int k = opts->nearest_k < 1 ? 1 : opts->nearest_k;
    if (op == IVJ_STREAM_NEAREST && k > 1024) return fail(IVJ_EINVAL, "nearest_k > 1024");
    DeviceGuard g(ctx->device);
    ivj_stream* st = new ivj_stream();
    st->ctx = ctx; st->opts = *opts; st->op = op; st->k = k; st->max_rows = max_batch_rows;
    ctx->streams.push_back(st);
    auto bail = [&](int rc) { ivj_stream_close(st); return rc; };
    {
        DevSide db;
        int rc = upload_side(ctx, build, db);
        if (rc != IVJ_OK) return bail(rc);
        const int general = op == IVJ_STREAM_NEAREST && !(k == 1 && opts->include_overlaps);
        rc = index_build(ctx, &db.s, opts, (op == IVJ_STREAM_COUNT || general) ? 1 : 0, &st->ix);
        if (rc != IVJ_OK) return bail(rc);
        hipError_t e = hipStreamSynchronize(ctx->stream);                      // the index copied what it needs: the upload may go
        if (e != hipSuccess) return bail(fail(IVJ_EHIP, std::string("sync(stream open): ") + hipGetErrorString(e)));
    }
    hipError_t e = hipStreamCreateWithFlags(&st->s_h2d, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st->s_d2h, hipStreamNonBlocking);
    const size_t col = align_up((size_t)max_batch_rows * 4);
    for (int s = 0; s < 3 && e == hipSuccess; ++s) {
        ivj_ctx::StreamBufs& cb = ctx->st_cache[s];                            // the staging of the previous session, if it is large enough
        if (cb.h_in && cb.d_in && cb.in_cap >= 3 * col) {
            st->slot[s].h_in = cb.h_in; st->slot[s].d_in = cb.d_in; st->slot[s].in_cap = cb.in_cap;
            st->slot[s].d_out = cb.d_out; st->slot[s].d_out_cap = cb.d_out_cap; st->slot[s].h_out = cb.h_out; st->slot[s].h_out_cap = cb.h_out_cap;
            cb = ivj_ctx::StreamBufs();
        } else {
            e = hipHostMalloc((void**)&st->slot[s].h_in, 3 * col, hipHostMallocDefault);
            if (e == hipSuccess) e = hipMalloc((void**)&st->slot[s].d_in, 3 * col);
            st->slot[s].in_cap = 3 * col;
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&st->slot[s].ev_h2d, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&st->slot[s].ev_join, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&st->slot[s].ev_d2h, hipEventDisableTiming);
    }
    if (e != hipSuccess) return bail(fail(IVJ_ENOMEM, std::string("stream slots: ") + hipGetErrorString(e)));
    *out = st;
    return IVJ_OK;
} IVJ_ABI_CATCH

int ivj_stream_submit(ivj_stream* st, const ivj_side* batch, ivj_stream_result* done) try {
    if (!st || !done) return fail(IVJ_EINVAL, "stream or done is NULL");
    IVJ_TRY(check_side(batch, "batch"));
    if (batch->n > st->max_rows) return fail(IVJ_EINVAL, "batch has more rows than max_batch_rows");
    if (batch->row_id) return fail(IVJ_EINVAL, "stream batches report rows inside the batch: row_id must be NULL");
    if (!st->ctx) return fail(IVJ_ESTATE, "the context of this streaming session was destroyed");
    return stream_turn(st, batch, done);
} IVJ_ABI_CATCH

int ivj_stream_flush(ivj_stream* st, ivj_stream_result* done) try {
    if (!st || !done) return fail(IVJ_EINVAL, "stream or done is NULL");
    if (!st->ctx) return fail(IVJ_ESTATE, "the context of this streaming session was destroyed");
    return stream_turn(st, nullptr, done);
} IVJ_ABI_CATCH

namespace {
// device / pinned resources of a streaming session; the staging goes back to the context's cache when it is still there
void stream_release(ivj_stream* st, bool keep_cache) {
    ivj_ctx* ctx = st->ctx;
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    if (st->s_h2d) (void)hipStreamSynchronize(st->s_h2d);
    if (st->s_d2h) (void)hipStreamSynchronize(st->s_d2h);
    (void)hipStreamSynchronize(ctx->stream);
    for (int s = 0; s < 3; ++s) {
        ivj_stream::Slot& S = st->slot[s];
        ivj_ctx::StreamBufs& cb = ctx->st_cache[s];                            // keep the (larger) staging for the next session
        if (keep_cache && S.h_in && S.d_in && S.in_cap >= cb.in_cap) {
            free_stream_bufs(cb);
            cb.h_in = S.h_in; cb.d_in = S.d_in; cb.in_cap = S.in_cap; cb.d_out = S.d_out; cb.d_out_cap = S.d_out_cap; cb.h_out = S.h_out; cb.h_out_cap = S.h_out_cap;
        } else {
            if (S.h_in) (void)hipHostFree(S.h_in);
            if (S.d_in) (void)hipFree(S.d_in);
            if (S.d_out) (void)hipFree(S.d_out);
            if (S.h_out) (void)hipHostFree(S.h_out);
        }
        if (S.ev_h2d) (void)hipEventDestroy(S.ev_h2d);
        if (S.ev_join) (void)hipEventDestroy(S.ev_join);
        if (S.ev_d2h) (void)hipEventDestroy(S.ev_d2h);
        S = ivj_stream::Slot();
    }
    if (st->s_h2d) (void)hipStreamDestroy(st->s_h2d);
    if (st->s_d2h) (void)hipStreamDestroy(st->s_d2h);
    st->s_h2d = st->s_d2h = nullptr;
    if (st->ix) ivj_index_free(st->ix);
    st->ix = nullptr;
    ctx->streams.erase(std::remove(ctx->streams.begin(), ctx->streams.end(), st), ctx->streams.end());
    st->ctx = nullptr;
}
}  // namespace

void ivj_stream_close(ivj_stream* st) {
    if (!st) return;
    stream_release(st, true);          // no-op when ivj_ctx_destroy already detached the session
    delete st;
}

// ---------------------------------------------------------------- memory helpers

int ivj_dev_alloc(ivj_ctx* ctx, int64_t bytes, void** out) try {
    if (!ctx || !out || bytes < 0) return fail(IVJ_EINVAL, "bad argument");
    DeviceGuard g(ctx->device);
    *out = nullptr;
    if (bytes == 0) return IVJ_OK;
    hipError_t e = hipMalloc(out, (size_t)bytes);
    if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return IVJ_OK;
} IVJ_ABI_CATCH
int ivj_dev_free(ivj_ctx* ctx, void* p) try {
    if (!ctx) return fail(IVJ_EINVAL, "ctx is NULL");
    if (!p) return IVJ_OK;
    DeviceGuard g(ctx->device);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(p));
    return IVJ_OK;
} IVJ_ABI_CATCH
int ivj_memcpy_h2d(ivj_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes) try {
    if (!ctx || bytes < 0) return fail(IVJ_EINVAL, "bad argument");
    if (bytes == 0) return IVJ_OK;
    DeviceGuard g(ctx->device);
    HostXfer copy(ctx->stream, &ctx->xfer);
    copy.h2d(dst_dev, src_host, (size_t)bytes);
    HIP_TRY(copy.finish());
    return IVJ_OK;
} IVJ_ABI_CATCH
int ivj_memcpy_d2h(ivj_ctx* ctx, void* dst_host, const void* src_dev, int64_t bytes) try {
    if (!ctx || bytes < 0) return fail(IVJ_EINVAL, "bad argument");
    if (bytes == 0) return IVJ_OK;
    DeviceGuard g(ctx->device);
    HostXfer copy(ctx->stream, &ctx->xfer);
    copy.d2h(dst_host, src_dev, (size_t)bytes);
    HIP_TRY(copy.finish());
    return IVJ_OK;
} IVJ_ABI_CATCH

}  // extern "C"

#include "host_frontdoor.hip.h"
#include "host_arrow_stream.hip.h"
