// host_comm.hip.h -- multi-GPU inside the library: communicator over RCCL, all-gatherv of result batches, and the sharded
// overlap whose exchange overlaps the join.  Part of the single translation unit ivjoin.hip; not a stand-alone header.
//
// One rank per GPU (one process per GPU, or one process holding one ivj_ctx per device).  Intervals on different contigs
// never interact, so there is no collective inside the join; the variable-length result batches are exchanged with an
// all-gatherv = ncclAllGather of the per-rank counts + ONE grouped batch of ncclSend / ncclRecv: every GPU pair talks over
// its own xGMI link (point-to-point fabric), not over a ring.
//
// RCCL is loaded on first use (dlopen of librccl.so.1): a host that never creates a communicator needs no RCCL at all,
// and a process that already carries an RCCL (PyTorch bundles one) shares that instance instead of loading a second.
#pragma once

#include <dlfcn.h>

#include <condition_variable>
#include <deque>
#include <mutex>

namespace {

// ---- the few RCCL entry points this file needs, by their public C signatures (rccl.h) ---------------------------------
struct RcclUniqueId { char internal[128]; };
typedef void* RcclComm;
enum { RCCL_INT8 = 0, RCCL_INT32 = 2, RCCL_INT64 = 4 };
struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
    int (*CommInitAll)(RcclComm*, int, const int*) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;

const RcclApi* rccl_api() {
    std::call_once(g_rccl_once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            g_rccl.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.handle) break;
        }
        if (!g_rccl.handle) { g_rccl.error = std::string("RCCL is not loadable: ") + (dlerror() ? dlerror() : "librccl.so.1 not found"); return; }
        bool ok = true;
        auto sym = [&](const char* nm) { void* p = dlsym(g_rccl.handle, nm); if (!p) { ok = false; g_rccl.error = std::string("RCCL lacks ") + nm; } return p; };
        g_rccl.GetUniqueId = (int (*)(RcclUniqueId*))sym("ncclGetUniqueId");
        g_rccl.CommInitRank = (int (*)(RcclComm*, int, RcclUniqueId, int))sym("ncclCommInitRank");
        g_rccl.CommInitAll = (int (*)(RcclComm*, int, const int*))sym("ncclCommInitAll");
        g_rccl.CommDestroy = (int (*)(RcclComm))sym("ncclCommDestroy");
        g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, RcclComm, hipStream_t))sym("ncclAllGather");
        g_rccl.Send = (int (*)(const void*, size_t, int, int, RcclComm, hipStream_t))sym("ncclSend");
        g_rccl.Recv = (int (*)(void*, size_t, int, int, RcclComm, hipStream_t))sym("ncclRecv");
        g_rccl.GroupStart = (int (*)())sym("ncclGroupStart");
        g_rccl.GroupEnd = (int (*)())sym("ncclGroupEnd");
        g_rccl.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        if (!ok) { dlclose(g_rccl.handle); g_rccl.handle = nullptr; }
    });
    return g_rccl.handle ? &g_rccl : nullptr;
}

#define RCCL_TRY(api, expr)                                                                                   \
    do {                                                                                                      \
        int _r = (expr);                                                                                      \
        if (_r != 0) return fail(IVJ_EHIP, std::string(#expr) + ": " + ((api)->GetErrorString ? (api)->GetErrorString(_r) : "RCCL error")); \
    } while (0)

}  // namespace

struct ivj_comm {
    ivj_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    RcclComm comm = nullptr;             // nullptr for a single-rank communicator (nothing to exchange, RCCL never touched)
    hipStream_t xstream = nullptr;       // the exchange runs on its own stream so that it overlaps the join
    long long* d_counts = nullptr;       // world + 1 int64 in HBM
    long long* h_counts = nullptr;       // pinned mirror
    int32_t* stage[2] = {nullptr, nullptr};   // result staging of the sharded overlap (2 x {probe rows | build rows}), kept between calls
    int64_t stage_cap = 0;               //   pairs per staging buffer
    int32_t* iota = nullptr;             // 0 .. iota_n - 1: row ids of a probe side that brings none (its chunks need absolute rows)
    int64_t iota_n = 0;
};

namespace {

int comm_finish_create(ivj_comm* c) {
    DeviceGuard g(c->ctx->device);
    HIP_TRY(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc((void**)&c->d_counts, (size_t)(c->world + 1) * 8));
    HIP_TRY(hipHostMalloc((void**)&c->h_counts, (size_t)(c->world + 1) * 8, hipHostMallocDefault));
    return IVJ_OK;
}

// counts[r] = n of rank r (all ranks), on the exchange stream; synchronises that stream
int comm_allgather_counts(ivj_comm* c, int64_t n_local, int64_t* counts) {
    if (c->world == 1) { counts[0] = n_local; return IVJ_OK; }
    const RcclApi* api = rccl_api();
    c->h_counts[c->world] = (long long)n_local;
    HIP_TRY(hipMemcpyAsync(c->d_counts + c->world, c->h_counts + c->world, 8, hipMemcpyHostToDevice, c->xstream));
    RCCL_TRY(api, api->AllGather(c->d_counts + c->world, c->d_counts, 1, RCCL_INT64, c->comm, c->xstream));
    HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counts, (size_t)c->world * 8, hipMemcpyDeviceToHost, c->xstream));
    HIP_TRY(hipStreamSynchronize(c->xstream));
    for (int r = 0; r < c->world; ++r) counts[r] = (int64_t)c->h_counts[r];
    return IVJ_OK;
}

// For every column: recv[k] + dst_off + (exclusive prefix of counts)[r] <- rank r's send[k][0 .. counts[r]) ; elem_bytes per element.
// One grouped batch of sends / receives on the exchange stream (own slice: device copy).  Does NOT synchronise.
int comm_exchange(ivj_comm* c, const void* const* send, void* const* recv, int n_cols, int elem_bytes, const int64_t* counts, int64_t dst_off) {
    std::vector<int64_t> off((size_t)c->world + 1, 0);
    for (int r = 0; r < c->world; ++r) off[r + 1] = off[r] + counts[r];
    const int64_t n_local = counts[c->rank];
    for (int k = 0; k < n_cols; ++k)
        if (n_local > 0)
            HIP_TRY(hipMemcpyAsync((char*)recv[k] + (size_t)(dst_off + off[c->rank]) * elem_bytes, send[k], (size_t)n_local * elem_bytes, hipMemcpyDeviceToDevice, c->xstream));
    if (c->world == 1) return IVJ_OK;
    const RcclApi* api = rccl_api();
    RCCL_TRY(api, api->GroupStart());
    for (int k = 0; k < n_cols; ++k) {
        for (int peer = 0; peer < c->world; ++peer) {
            if (peer == c->rank) continue;
            if (n_local > 0) RCCL_TRY(api, api->Send(send[k], (size_t)n_local * elem_bytes, RCCL_INT8, peer, c->comm, c->xstream));
            if (counts[peer] > 0)
                RCCL_TRY(api, api->Recv((char*)recv[k] + (size_t)(dst_off + off[peer]) * elem_bytes, (size_t)counts[peer] * elem_bytes, RCCL_INT8, peer, c->comm, c->xstream));
        }
    }
    RCCL_TRY(api, api->GroupEnd());
    return IVJ_OK;
}


__global__ void k_iota(int32_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)i;
}

// Sharded pb.overlap whose exchange overlaps the join.  This rank's probe rows are cut into n_chunks contiguous chunks (the
// same number on every rank); chunk i is joined (fused single pass) into a staging buffer while a helper thread exchanges
// chunk i - 1: ncclAllGather of the chunk's per-rank counts, then the grouped send / receive batch straight into the caller's
// result columns.  Result layout: chunk after chunk, inside a chunk rank after rank (the reference leaves the row order of
// pb.overlap unspecified; the pairs of one probe row stay contiguous).  Every rank ends up with every pair.
int overlap_allgather(ivj_comm* c, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int n_chunks, int32_t* out_p, int32_t* out_b,
                      int64_t capacity, int64_t* n_total, int64_t* n_local_out) {
    ivj_ctx* ctx = c->ctx;
    const int64_t n = probe->n;
    *n_total = 0;
    if (n_local_out) *n_local_out = 0;
    const int32_t* row_id = probe->row_id;
    if (!row_id && n > 0 && n_chunks > 1) {
        if (c->iota_n < n) {
            if (c->iota) HIP_TRY(hipFree(c->iota));
            c->iota = nullptr; c->iota_n = 0;
            HIP_TRY(hipMalloc((void**)&c->iota, (size_t)n * 4));
            hipLaunchKernelGGL(k_iota, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, c->iota, n);
            HIP_TRY(hipGetLastError());
            c->iota_n = n;
        }
        row_id = c->iota;
    }
    auto ensure_stage = [&](int64_t cap) -> int {
        if (cap <= c->stage_cap) return IVJ_OK;
        for (auto& b : c->stage) { if (b) HIP_TRY(hipFree(b)); b = nullptr; }
        c->stage_cap = 0;
        for (auto& b : c->stage) HIP_TRY(hipMalloc((void**)&b, (size_t)cap * 8));
        c->stage_cap = cap;
        return IVJ_OK;
    };
    {
        int64_t guess = capacity / ((int64_t)c->world * n_chunks);
        guess += guess / 2 + (1 << 16);
        IVJ_TRY(ensure_stage(guess < capacity + 1 ? guess : capacity + 1));
    }
    // exchange thread: one job per chunk, in order
    struct Job { int chunk; int64_t n; int buf; };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> q;
    bool done[2] = {true, true};                      // staging buffer free again
    int x_rc = IVJ_OK;
    std::string x_err;
    int64_t dst_off = 0, local_sum = 0;
    bool stop = false;
    std::thread xt([&] {
        (void)hipSetDevice(ctx->device);
        std::vector<int64_t> counts((size_t)c->world);
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                j = q.front(); q.pop_front();
            }
            int rc = x_rc;
            if (rc == IVJ_OK) rc = comm_allgather_counts(c, j.n, counts.data());        // every rank takes part in every chunk's collectives
            if (rc == IVJ_OK) {
                int64_t tot = 0;
                for (int r = 0; r < c->world; ++r) tot += counts[r];
                if (dst_off + tot > capacity) { rc = fail(IVJ_ECAPACITY, "output capacity " + std::to_string(capacity) + " < " + std::to_string(dst_off + tot) + " pairs after chunk " + std::to_string(j.chunk)); dst_off += tot; }
                else {
                    const void* send[2] = {c->stage[j.buf], c->stage[j.buf] + c->stage_cap};
                    void* recv[2] = {out_p, out_b};
                    rc = comm_exchange(c, send, recv, 2, 4, counts.data(), dst_off);
                    if (rc == IVJ_OK && hipStreamSynchronize(c->xstream) != hipSuccess) rc = fail(IVJ_EHIP, "exchange stream synchronize failed");
                    dst_off += tot;
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (rc != IVJ_OK && x_rc == IVJ_OK) { x_rc = rc; x_err = g_err; }
                done[j.buf] = true;
            }
            cv.notify_all();
        }
    });
    int rc = IVJ_OK;
    for (int i = 0; i < n_chunks && rc == IVJ_OK; ++i) {
        const int64_t lo = n * i / n_chunks, hi = n * (i + 1) / n_chunks;
        const int buf = i & 1;
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done[buf]; }); }
        int64_t got = 0;
        if (hi > lo && ix->n > 0) {
            ivj_side sub{probe->contig + lo, probe->start + lo, probe->end + lo, hi - lo, row_id ? row_id + lo : nullptr};
            rc = overlap_fused(ctx, ix, &sub, opts, c->stage[buf], c->stage[buf] + c->stage_cap, c->stage_cap, &got);
            if (rc == IVJ_ECAPACITY) {
                // the chunk's pairs did not fit the staging: drain the exchange, grow both buffers, redo the chunk
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done[0] && done[1]; }); }
                rc = ensure_stage(got + got / 8 + 1024);
                if (rc == IVJ_OK) rc = overlap_fused(ctx, ix, &sub, opts, c->stage[buf], c->stage[buf] + c->stage_cap, c->stage_cap, &got);
            }
        }
        if (rc != IVJ_OK) got = 0;                                   // keep the collective sequence of the other ranks alive
        local_sum += got;
        { std::lock_guard<std::mutex> lk(mu); done[buf] = false; q.push_back(Job{i, got, buf}); }
        cv.notify_all();
    }
    const std::string main_err = g_err;
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    xt.join();
    *n_total = dst_off;
    if (n_local_out) *n_local_out = local_sum;
    if (rc != IVJ_OK) { g_err = main_err; return rc; }
    if (x_rc != IVJ_OK) { g_err = x_err; return x_rc; }
    return IVJ_OK;
}

}  // namespace
