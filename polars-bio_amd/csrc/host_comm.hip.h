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

#include <chrono>
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

namespace {
// In-process transport for the ranks of ONE process (ivj_comm_create_local): a rendezvous in host memory + device copies
// issued by the RECEIVING rank.  It carries the same protocol as the RCCL transport (same calls in the same order, the same
// blocking behaviour: a rank that skips a collective leaves its peers waiting -- here until LOOP_TIMEOUT_S, then IVJ_EHIP),
// and it is what lets two ranks share one device, which RCCL refuses: a 1-GPU box runs the world-2 protocol of the sharded
// overlap with it.  Chosen by ivj_comm_create_local when two contexts share a device or IVJ_COMM_LOOPBACK=1 is set.
struct LoopGroup {
    std::mutex mu;
    std::condition_variable cv;
    int world = 0, arrived = 0, refs = 0;
    uint64_t gen = 0;
    std::vector<int64_t> vals;                        // world * 2
    std::vector<const void*> send;                    // world * LOOP_MAX_COLS
    std::vector<int64_t> cnt;                         // world
    static constexpr int MAX_COLS = 8;
    // -> false on timeout
    bool barrier(int timeout_s) {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t g = gen;
        if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); return true; }
        if (cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return gen != g; })) return true;
        --arrived;                                   // give up: the barrier stays consistent for whoever comes later
        return false;
    }
};
int loop_timeout_s() { const char* ev = std::getenv("IVJ_COMM_LOOPBACK_TIMEOUT"); const int t = ev ? std::atoi(ev) : 0; return t > 0 ? t : 120; }
}  // namespace

struct ivj_comm {
    ivj_ctx* ctx = nullptr;              // nullptr once the context was destroyed under the communicator (calls then fail with IVJ_ESTATE)
    int device = 0;
    int rank = 0, world = 1;
    RcclComm comm = nullptr;             // nullptr for a single-rank communicator (nothing to exchange, RCCL never touched)
    bool self_rccl = false;              // IVJ_COMM_NO_SHORTCUT=1 at creation: a world-1 communicator is a REAL RCCL communicator and this rank's
                                         //   own slice travels through ncclSend / ncclRecv to itself -- the way a 1-GPU box executes the RCCL branch
    LoopGroup* loop = nullptr;           // in-process transport instead of RCCL (shared by the communicators of one create_local call)
    hipStream_t xstream = nullptr;       // the exchange runs on its own stream so that it overlaps the join
    long long* d_counts = nullptr;       // 2 * (world + 1) int64 in HBM: gathered values (<= 2 per rank), then this rank's send slots
    long long* h_counts = nullptr;       // pinned mirror
    int32_t* stage[2] = {nullptr, nullptr};   // result staging of the sharded overlap (2 x {probe rows | build rows}), kept between calls
    int64_t stage_cap = 0;               //   pairs per staging buffer
    int32_t* iota = nullptr;             // 0 .. iota_n - 1: row ids of a probe side that brings none (its chunks need absolute rows)
    int64_t iota_n = 0;
    char* pp_buf = nullptr;              // scratch of the per-probe exchange (count_overlaps / nearest): local results, send and receive columns
    size_t pp_cap = 0;
    long long pp_scatter_fallbacks = 0;  // per-probe exchanges whose senders were not ascending (scatter form instead of the merge)
};

namespace {

// the context goes away under a live communicator: the device resources that live on the context's device stay with the
// communicator (ivj_comm_destroy releases them), only the way back to the context is cut
void comm_detach(ivj_comm* c) {
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);
    c->ctx = nullptr;
}

int comm_finish_create(ivj_comm* c) {
    c->device = c->ctx->device;
    c->ctx->comms.push_back(c);
    DeviceGuard g(c->device);
    HIP_TRY(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc((void**)&c->d_counts, (size_t)(c->world + 1) * 16));
    HIP_TRY(hipHostMalloc((void**)&c->h_counts, (size_t)(c->world + 1) * 16, hipHostMallocDefault));
    return IVJ_OK;
}

// For every column: recv[k] + dst_off + (exclusive prefix of counts)[r] <- rank r's send[k][0 .. counts[r]) ; elem_bytes per element.
// One grouped batch of sends / receives on the exchange stream (own slice: device copy).  Does NOT synchronise.
// own_copy = false: this rank's own slice is NOT copied into recv (the caller reads it where it lies); with the self-RCCL mode it travels like a peer's.
int comm_exchange_v(ivj_comm* c, const void* const* send, void* const* recv, int n_cols, const int* elem_bytes_v, const int64_t* counts, int64_t dst_off, bool own_copy = true) {
    std::vector<int64_t> off((size_t)c->world + 1, 0);
    for (int r = 0; r < c->world; ++r) off[r + 1] = off[r] + counts[r];
    const int64_t n_local = counts[c->rank];
    const bool self_rccl = c->self_rccl && c->comm;   // own slice through RCCL too (send / receive to self inside the group)
    for (int k = 0; k < n_cols && !self_rccl && own_copy; ++k) {
        const int elem_bytes = elem_bytes_v[k];
        if (n_local > 0)
            HIP_TRY(hipMemcpyAsync((char*)recv[k] + (size_t)(dst_off + off[c->rank]) * elem_bytes, send[k], (size_t)n_local * elem_bytes, hipMemcpyDeviceToDevice, c->xstream));
    }
    if (c->world == 1 && !self_rccl) return IVJ_OK;
    if (c->loop) {
        LoopGroup* L = c->loop;
        if (n_cols > LoopGroup::MAX_COLS) return fail(IVJ_EINVAL, "loopback transport: too many columns");
        { std::lock_guard<std::mutex> lk(L->mu); for (int k = 0; k < n_cols; ++k) L->send[(size_t)c->rank * LoopGroup::MAX_COLS + k] = send[k]; }
        if (!L->barrier(loop_timeout_s())) return fail(IVJ_EHIP, "loopback transport: a rank did not reach the exchange (timeout)");
        hipError_t e = hipSuccess;
        for (int k = 0; k < n_cols && e == hipSuccess; ++k)
            for (int peer = 0; peer < c->world && e == hipSuccess; ++peer)
                if (peer != c->rank && counts[peer] > 0)
                    e = hipMemcpyAsync((char*)recv[k] + (size_t)(dst_off + off[peer]) * elem_bytes_v[k], L->send[(size_t)peer * LoopGroup::MAX_COLS + k],
                                       (size_t)counts[peer] * elem_bytes_v[k], hipMemcpyDefault, c->xstream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->xstream);
        if (!L->barrier(loop_timeout_s())) return fail(IVJ_EHIP, "loopback transport: a rank did not finish the exchange (timeout)");   // the peers' send buffers are free again
        if (e != hipSuccess) return fail(IVJ_EHIP, std::string("loopback transport copy: ") + hipGetErrorString(e));
        return IVJ_OK;
    }
    const RcclApi* api = rccl_api();
    RCCL_TRY(api, api->GroupStart());
    for (int k = 0; k < n_cols; ++k) {
        const int elem_bytes = elem_bytes_v[k];
        for (int peer = 0; peer < c->world; ++peer) {
            if (peer == c->rank && !self_rccl) continue;
            if (n_local > 0) RCCL_TRY(api, api->Send(send[k], (size_t)n_local * elem_bytes, RCCL_INT8, peer, c->comm, c->xstream));
            if (counts[peer] > 0)
                RCCL_TRY(api, api->Recv((char*)recv[k] + (size_t)(dst_off + off[peer]) * elem_bytes, (size_t)counts[peer] * elem_bytes, RCCL_INT8, peer, c->comm, c->xstream));
        }
    }
    RCCL_TRY(api, api->GroupEnd());
    return IVJ_OK;
}
int comm_exchange(ivj_comm* c, const void* const* send, void* const* recv, int n_cols, int elem_bytes, const int64_t* counts, int64_t dst_off) {
    if (n_cols > LoopGroup::MAX_COLS) return fail(IVJ_EINVAL, "all-gatherv: too many columns");
    int w[LoopGroup::MAX_COLS];
    for (int k = 0; k < n_cols; ++k) w[k] = elem_bytes;
    return comm_exchange_v(c, send, recv, n_cols, w, counts, dst_off);
}


__global__ void k_iota(int32_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)i;
}

// (values[0 .. nv) of every rank) -> all[r * nv + v], on the exchange stream; synchronises that stream.  nv <= 2.
int comm_allgather_i64(ivj_comm* c, const int64_t* vals, int nv, int64_t* all) {
    if (c->world == 1 && !(c->self_rccl && c->comm)) { for (int v = 0; v < nv; ++v) all[v] = vals[v]; return IVJ_OK; }
    if (c->loop) {
        LoopGroup* L = c->loop;
        { std::lock_guard<std::mutex> lk(L->mu); for (int v = 0; v < nv; ++v) L->vals[(size_t)c->rank * 2 + v] = vals[v]; }
        if (!L->barrier(loop_timeout_s())) return fail(IVJ_EHIP, "loopback transport: a rank did not reach the count all-gather (timeout)");
        { std::lock_guard<std::mutex> lk(L->mu); for (int r = 0; r < c->world; ++r) for (int v = 0; v < nv; ++v) all[(size_t)r * nv + v] = L->vals[(size_t)r * 2 + v]; }
        if (!L->barrier(loop_timeout_s())) return fail(IVJ_EHIP, "loopback transport: a rank did not leave the count all-gather (timeout)");
        return IVJ_OK;
    }
    const RcclApi* api = rccl_api();
    long long* h_send = c->h_counts + (size_t)c->world * 2;
    long long* d_send = c->d_counts + (size_t)c->world * 2;
    for (int v = 0; v < nv; ++v) h_send[v] = (long long)vals[v];
    HIP_TRY(hipMemcpyAsync(d_send, h_send, (size_t)nv * 8, hipMemcpyHostToDevice, c->xstream));
    RCCL_TRY(api, api->AllGather(d_send, c->d_counts, (size_t)nv, RCCL_INT64, c->comm, c->xstream));
    HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counts, (size_t)c->world * nv * 8, hipMemcpyDeviceToHost, c->xstream));
    HIP_TRY(hipStreamSynchronize(c->xstream));
    for (int i = 0; i < c->world * nv; ++i) all[i] = (int64_t)c->h_counts[i];
    return IVJ_OK;
}

// counts[r] = n of rank r (all ranks), on the exchange stream; synchronises that stream
int comm_allgather_counts(ivj_comm* c, int64_t n_local, int64_t* counts) { return comm_allgather_i64(c, &n_local, 1, counts); }

// IVJ_FAULT_ALLGATHER="<rank>:<chunk>" (test knob): the join of that chunk on that rank reports a failure instead of running.
bool fault_injected(int rank, int chunk) {
    const char* ev = std::getenv("IVJ_FAULT_ALLGATHER");
    if (!ev || !*ev) return false;
    int r = -1, ch = -1;
    if (std::sscanf(ev, "%d:%d", &r, &ch) != 2) return false;
    return r == rank && ch == chunk;
}

// Sharded pb.overlap whose exchange overlaps the join.  This rank's probe rows are cut into n_chunks contiguous chunks (the
// same number on every rank); chunk i is joined (fused single pass) into a staging buffer while a helper thread exchanges
// chunk i - 1: ncclAllGather of the chunk's per-rank {pair count | failure mark, room left in the caller's columns}, then the
// grouped send / receive batch straight into the caller's result columns.  Result layout: chunk after chunk, inside a chunk
// rank after rank (the reference leaves the row order of pb.overlap unspecified; the pairs of one probe row stay contiguous).
// Every rank ends up with every pair.
//
// Failure protocol -- no rank is ever left waiting in a collective:
//   * EVERY rank issues the count all-gather of EVERY chunk, whatever happened to it before.  A rank whose join failed
//     submits the failed chunk and all later ones with the mark -1 (it does not join them) and returns its own error at the end;
//     the other ranks read the mark as "0 pairs from that rank", finish, and return IVJ_EPEER.
//   * the decision to move a chunk's pairs is taken from the GATHERED values only (pairs of the chunk against the smallest
//     room any rank has left), so all ranks take it alike: once one rank's columns are too small, no rank sends or receives
//     any more, the remaining chunks are still counted, and every rank returns IVJ_ECAPACITY with *n_total = the capacity
//     the call needs.
//   * only an error of the collective itself (RCCL / the exchange stream) ends this rank's participation: the fabric is
//     gone then and nothing can be promised to the peers.
int overlap_allgather(ivj_comm* c, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int n_chunks, int32_t* out_p, int32_t* out_b,
                      int64_t capacity, int64_t* n_total, int64_t* n_local_out) {
    ivj_ctx* ctx = c->ctx;
    const int64_t n = probe->n;
    *n_total = 0;
    if (n_local_out) *n_local_out = 0;
    int rc = IVJ_OK;                                   // first failure of this rank's own work (joins, allocations)
    std::string main_err;
    auto note = [&](int r) { if (r != IVJ_OK && rc == IVJ_OK) { rc = r; main_err = g_err; } };
    const int32_t* row_id = probe->row_id;
    if (!row_id && n > 0 && n_chunks > 1) {
        if (c->iota_n < n) {
            if (c->iota) (void)hipFree(c->iota);
            c->iota = nullptr; c->iota_n = 0;
            if (hipMalloc((void**)&c->iota, (size_t)n * 4) != hipSuccess) note(fail(IVJ_ENOMEM, "row ids of the chunks: hipMalloc failed"));
            else {
                hipLaunchKernelGGL(k_iota, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, c->iota, n);
                if (hipGetLastError() != hipSuccess) note(fail(IVJ_EHIP, "k_iota launch failed"));
                else c->iota_n = n;
            }
        }
        row_id = c->iota;
    }
    auto ensure_stage = [&](int64_t cap) -> int {
        if (cap <= c->stage_cap) return IVJ_OK;
        for (auto& b : c->stage) { if (b) (void)hipFree(b); b = nullptr; }
        c->stage_cap = 0;
        for (auto& b : c->stage)
            if (hipMalloc((void**)&b, (size_t)cap * 8) != hipSuccess) return fail(IVJ_ENOMEM, "staging of the sharded overlap: hipMalloc of " + std::to_string(cap * 8) + " bytes failed");
        c->stage_cap = cap;
        return IVJ_OK;
    };
    if (rc == IVJ_OK) {
        int64_t guess = capacity / ((int64_t)c->world * n_chunks);
        guess += guess / 2 + (1 << 16);
        note(ensure_stage(guess < capacity + 1 ? guess : capacity + 1));
    }
    // exchange thread: one job per chunk, in order
    struct Job { int chunk; int64_t n; int buf; };      // n = pairs of the chunk on this rank, -1: this rank has failed
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> q;
    bool done[2] = {true, true};                      // staging buffer free again
    int x_rc = IVJ_OK;                                // error of the collectives themselves
    std::string x_err;
    bool overflow = false, peer_failed = false;
    int failed_peer = -1, failed_chunk = -1;
    int64_t dst_off = 0, local_sum = 0;
    bool stop = false;
    std::thread xt([&] {
        (void)hipSetDevice(ctx->device);
        std::vector<int64_t> all((size_t)c->world * 2), counts((size_t)c->world);
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                j = q.front(); q.pop_front();
            }
            int xr = IVJ_OK;
            if (x_rc == IVJ_OK) {
                const int64_t mine[2] = {j.n, capacity - dst_off};
                xr = comm_allgather_i64(c, mine, 2, all.data());             // every rank, every chunk
                if (xr == IVJ_OK) {
                    int64_t tot = 0, room = capacity - dst_off;
                    for (int r = 0; r < c->world; ++r) {
                        const int64_t nr = all[(size_t)r * 2];
                        if (nr < 0 && r != c->rank && !peer_failed) { peer_failed = true; failed_peer = r; failed_chunk = j.chunk; }
                        counts[r] = nr < 0 ? 0 : nr;
                        tot += counts[r];
                        if (all[(size_t)r * 2 + 1] < room) room = all[(size_t)r * 2 + 1];
                    }
                    if (tot > room) overflow = true;                         // sticky, and the same on every rank
                    if (!overflow) {
                        const void* send[2] = {c->stage[j.buf], c->stage[j.buf] + c->stage_cap};
                        void* recv[2] = {out_p, out_b};
                        xr = comm_exchange(c, send, recv, 2, 4, counts.data(), dst_off);
                        if (xr == IVJ_OK && hipStreamSynchronize(c->xstream) != hipSuccess) xr = fail(IVJ_EHIP, "exchange stream synchronize failed");
                    }
                    dst_off += tot;
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (xr != IVJ_OK && x_rc == IVJ_OK) { x_rc = xr; x_err = g_err; }
                done[j.buf] = true;
            }
            cv.notify_all();
        }
    });
    struct Joiner {                                    // the helper is joined on every way out of this frame (also an exception's)
        std::thread& t; std::mutex& mu; std::condition_variable& cv; bool& stop;
        ~Joiner() { if (t.joinable()) { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); t.join(); } }
    } joiner{xt, mu, cv, stop};
    for (int i = 0; i < n_chunks; ++i) {
        const int64_t lo = n * i / n_chunks, hi = n * (i + 1) / n_chunks;
        const int buf = i & 1;
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done[buf]; }); }
        int64_t got = 0;
        if (rc == IVJ_OK && fault_injected(c->rank, i)) note(fail(IVJ_EHIP, "injected fault (IVJ_FAULT_ALLGATHER) in chunk " + std::to_string(i)));
        if (rc == IVJ_OK && hi > lo && ix->n > 0) {
            ivj_side sub{probe->contig + lo, probe->start + lo, probe->end + lo, hi - lo, row_id ? row_id + lo : nullptr};
            int r = overlap_fused(ctx, ix, &sub, opts, c->stage[buf], c->stage[buf] + c->stage_cap, c->stage_cap, &got);
            if (r == IVJ_ECAPACITY) {
                // the chunk's pairs did not fit the staging: drain the exchange, grow both buffers, redo the chunk
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done[0] && done[1]; }); }
                r = ensure_stage(got + got / 8 + 1024);
                if (r == IVJ_OK) r = overlap_fused(ctx, ix, &sub, opts, c->stage[buf], c->stage[buf] + c->stage_cap, c->stage_cap, &got);
            }
            note(r);
        }
        if (rc != IVJ_OK) got = -1;                                  // failure mark: this and every later chunk still reach the collective
        else local_sum += got;
        { std::lock_guard<std::mutex> lk(mu); done[buf] = false; q.push_back(Job{i, got, buf}); }
        cv.notify_all();
    }
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    xt.join();
    *n_total = dst_off;
    if (n_local_out) *n_local_out = local_sum;
    if (rc != IVJ_OK) { g_err = main_err; return rc; }
    if (x_rc != IVJ_OK) { g_err = x_err; return x_rc; }
    if (peer_failed) return fail(IVJ_EPEER, "rank " + std::to_string(failed_peer) + " failed in chunk " + std::to_string(failed_chunk) + " of the sharded overlap; its pairs from there on are missing");
    if (overflow) return fail(IVJ_ECAPACITY, "output capacity " + std::to_string(capacity) + " < " + std::to_string(dst_off) + " pairs (on some rank): nothing was exchanged past the first chunk that did not fit");
    return IVJ_OK;
}

// ---- count_overlaps / nearest of a shard + the exchange of the PER-PROBE results ---------------------------------
// Every probe row lives on exactly one rank (contig sharding; global row in ivj_side.row_id), its result has a fixed width, and every
// rank wants the full-length columns in the ORIGINAL probe order.
//   wire, count_overlaps: {row int32, count int32}  (a count is bounded by the build rows, < 2^31; widened to int64 by the receiver)
//   wire, nearest:        {row int32, k build rows int32, k distances int64, n_found int32}
// Rows no rank reports keep the defaults (count 0; build row -1, distance -1, n_found 0).
//
// Round 6 -- the receiver MERGES instead of scattering.  A shard made by ivj_host_shard (or any host that keeps df1's order) lists its
// global rows in ASCENDING order, so the rows of one output tile [t0, t0 + PP_TILE) are ONE contiguous segment of every sender's columns:
// k_pp_tile_offsets finds the segment bounds (one bound search per (sender, tile), all in parallel), k_pp_merge_* reads the segments
// coalesced, places the values by row in LDS and writes the tile coalesced with the defaults filled in -- no fill pass, no random 8-byte
// store per row (round 5: 200 M of them for config 5, 8.4 of the call's 10.6 ms), no pack pass (the shard's kernel writes the wire columns
// itself, the row column on the wire IS probe.row_id), and this rank's own slice is read where it lies instead of being copied.
// Exactness does not rest on the order: every tile checks that what it placed belongs to it and that no row came twice, the placed rows
// are counted, and a call whose senders are not ascending falls back to the scatter kernels (round-5 form: fill + one store per row).
constexpr int PP_TILE = 4096;
constexpr int PP_THREADS = 256;
constexpr int PP_MAX_WORLD = 64;       // senders the merge kernels take (their table is a kernel argument); larger worlds scatter

struct PpSrc {
    int world, own;                     // own: the rank whose columns are read from the own_* pointers (-1: every slice lies in the receive columns)
    long long off[PP_MAX_WORLD + 1];    // exclusive prefix of the per-rank row counts = position of rank r's slice in the receive columns
};
// flags[0] bit 0: a row outside [0, n_total) / outside the tile its segment belongs to (senders not ascending)   bit 1: a row reported twice
// flags[2..3]: rows placed (uint64)

__device__ __forceinline__ uint32_t pp_lower_bound(const int32_t* __restrict__ rows, uint32_t n, long long key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t m = lo + ((hi - lo) >> 1);
        if ((long long)rows[m] < key) lo = m + 1; else hi = m;
    }
    return lo;
}

// toff[r * (ntiles + 1) + t] = first position of sender r whose row is >= t * PP_TILE   (rows == nullptr: the identity, world 1 only)
__global__ void k_pp_tile_offsets(PpSrc S, const int32_t* __restrict__ recv_rows, const int32_t* __restrict__ own_rows, int64_t ntiles,
                                  uint32_t* __restrict__ toff) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)S.world * (ntiles + 1)) return;
    const int r = (int)(i / (ntiles + 1));
    const int64_t t = i - (int64_t)r * (ntiles + 1);
    const uint32_t n_r = (uint32_t)(S.off[r + 1] - S.off[r]);
    const int32_t* rows = r == S.own ? own_rows : recv_rows + S.off[r];
    const long long key = (long long)t * PP_TILE;
    toff[i] = (r == S.own && !own_rows) ? (uint32_t)(key < (long long)n_r ? key : (long long)n_r) : pp_lower_bound(rows, n_r, key);
}

__global__ __launch_bounds__(PP_THREADS) void k_pp_merge_count(PpSrc S, const int32_t* __restrict__ recv_rows, const int32_t* __restrict__ own_rows,
                                                               const int32_t* __restrict__ recv_cnt, const int32_t* __restrict__ own_cnt,
                                                               int64_t n_total, int64_t ntiles, const uint32_t* __restrict__ toff,
                                                               long long* __restrict__ out, unsigned int* __restrict__ flags) {
    __shared__ int tile[PP_TILE];
    __shared__ unsigned int l_placed, l_bad;
    const int64_t t = blockIdx.x, t0 = t * PP_TILE;
    for (int i = threadIdx.x; i < PP_TILE; i += PP_THREADS) tile[i] = -1;
    if (threadIdx.x == 0) { l_placed = 0; l_bad = 0; }
    __syncthreads();
    unsigned int placed = 0, bad = 0;
    for (int r = 0; r < S.world; ++r) {
        const uint32_t a = toff[(int64_t)r * (ntiles + 1) + t];
        uint32_t b = toff[(int64_t)r * (ntiles + 1) + t + 1];
        if (b < a) b = a;
        if (b - a > (uint32_t)PP_TILE) { bad |= 1u; b = a + PP_TILE; }
        const bool own = r == S.own;
        const int32_t* rows = own ? own_rows : recv_rows + S.off[r];
        const int32_t* cnt = own ? own_cnt : recv_cnt + S.off[r];
        for (uint32_t j = a + threadIdx.x; j < b; j += PP_THREADS) {
            const long long g = rows ? (long long)__builtin_nontemporal_load(rows + j) : (long long)j;
            const int v = __builtin_nontemporal_load(cnt + j);
            const long long slot = g - t0;
            if (slot < 0 || slot >= PP_TILE || g >= n_total || v < 0) { bad |= 1u; continue; }
            if (atomicExch(&tile[slot], v) != -1) bad |= 2u;
            ++placed;
        }
    }
    if (placed) atomicAdd(&l_placed, placed);
    if (bad) atomicOr(&l_bad, bad);
    __syncthreads();
    // the tile, coalesced, defaults filled in: two int64 per 16-byte store
    typedef long long v2ll __attribute__((ext_vector_type(2)));
    if (t0 + PP_TILE <= n_total && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {
        for (int i = 2 * threadIdx.x; i < PP_TILE; i += 2 * PP_THREADS) {
            const int2 c = *reinterpret_cast<const int2*>(&tile[i]);
            v2ll v; v.x = c.x < 0 ? 0 : c.x; v.y = c.y < 0 ? 0 : c.y;
            __builtin_nontemporal_store(v, reinterpret_cast<v2ll*>(out + t0 + i));
        }
    } else {
        for (int i = threadIdx.x; i < PP_TILE; i += PP_THREADS)
            if (t0 + i < n_total) out[t0 + i] = tile[i] < 0 ? 0 : tile[i];
    }
    if (threadIdx.x == 0) {
        if (l_placed) atomicAdd(reinterpret_cast<unsigned long long*>(flags + 2), (unsigned long long)l_placed);
        if (l_bad) atomicOr(flags, l_bad);
    }
}

// nearest: the tile holds a REFERENCE per output row (sender << 13 | position inside the sender's segment); the k build rows, k distances
// and n_found of a row are fetched through it at write-out, where consecutive lanes read consecutive positions of (at most `world`) segments
__global__ __launch_bounds__(PP_THREADS) void k_pp_merge_nearest(PpSrc S, const int32_t* __restrict__ recv_rows, const int32_t* __restrict__ own_rows,
                                                                 const int32_t* __restrict__ recv_idx, const int32_t* __restrict__ own_idx,
                                                                 const long long* __restrict__ recv_dist, const long long* __restrict__ own_dist,
                                                                 const int32_t* __restrict__ recv_nf, const int32_t* __restrict__ own_nf, int k,
                                                                 int64_t n_total, int64_t ntiles, const uint32_t* __restrict__ toff,
                                                                 int32_t* __restrict__ o_idx, long long* __restrict__ o_dist, int32_t* __restrict__ o_nf,
                                                                 unsigned int* __restrict__ flags) {
    __shared__ uint32_t tile[PP_TILE];
    __shared__ uint32_t seg_a[PP_MAX_WORLD];
    __shared__ unsigned int l_placed, l_bad;
    const int64_t t = blockIdx.x, t0 = t * PP_TILE;
    for (int i = threadIdx.x; i < PP_TILE; i += PP_THREADS) tile[i] = 0xffffffffu;
    if (threadIdx.x < S.world) seg_a[threadIdx.x] = toff[(int64_t)threadIdx.x * (ntiles + 1) + t];
    if (threadIdx.x == 0) { l_placed = 0; l_bad = 0; }
    __syncthreads();
    unsigned int placed = 0, bad = 0;
    for (int r = 0; r < S.world; ++r) {
        const uint32_t a = seg_a[r];
        uint32_t b = toff[(int64_t)r * (ntiles + 1) + t + 1];
        if (b < a) b = a;
        if (b - a > (uint32_t)PP_TILE) { bad |= 1u; b = a + PP_TILE; }
        const int32_t* rows = r == S.own ? own_rows : recv_rows + S.off[r];
        for (uint32_t j = a + threadIdx.x; j < b; j += PP_THREADS) {
            const long long g = rows ? (long long)__builtin_nontemporal_load(rows + j) : (long long)j;
            const long long slot = g - t0;
            if (slot < 0 || slot >= PP_TILE || g >= n_total) { bad |= 1u; continue; }
            if (atomicExch(&tile[slot], ((uint32_t)r << 13) | (j - a)) != 0xffffffffu) bad |= 2u;
            ++placed;
        }
    }
    if (placed) atomicAdd(&l_placed, placed);
    if (bad) atomicOr(&l_bad, bad);
    __syncthreads();
    for (int i = threadIdx.x; i < PP_TILE; i += PP_THREADS) {
        const int64_t g = t0 + i;
        if (g >= n_total) break;
        const uint32_t ref = tile[i];
        if (ref == 0xffffffffu) {
            for (int q = 0; q < k; ++q) { o_idx[g * k + q] = -1; o_dist[g * k + q] = -1; }
            o_nf[g] = 0;
        } else {
            const int r = (int)(ref >> 13);
            const int64_t j = (int64_t)seg_a[r] + (ref & 8191u);
            const bool own = r == S.own;
            const int32_t* s_idx = own ? own_idx : recv_idx + S.off[r] * k;
            const long long* s_dist = own ? own_dist : recv_dist + S.off[r] * k;
            const int32_t* s_nf = own ? own_nf : recv_nf + S.off[r];
            for (int q = 0; q < k; ++q) { o_idx[g * k + q] = s_idx[j * k + q]; o_dist[g * k + q] = s_dist[j * k + q]; }
            o_nf[g] = s_nf[j];
        }
    }
    if (threadIdx.x == 0) {
        if (l_placed) atomicAdd(reinterpret_cast<unsigned long long*>(flags + 2), (unsigned long long)l_placed);
        if (l_bad) atomicOr(flags, l_bad);
    }
}

// the scatter form (senders in any order): one store per reported row into columns that hold the defaults already
__global__ void k_pp_rows(const int32_t* __restrict__ row_id, int64_t n, int32_t* __restrict__ o_row) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o_row[i] = row_id ? row_id[i] : (int32_t)i;
}
__global__ void k_pp_scatter_count(const int32_t* __restrict__ row, const int32_t* __restrict__ cnt, int64_t n, int64_t n_out, long long* __restrict__ out,
                                   unsigned int* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = row ? (int64_t)row[i] : i;
    if ((uint64_t)r >= (uint64_t)n_out) { atomicOr(bad, 1u); return; }
    out[r] = (long long)cnt[i];
}
__global__ void k_pp_scatter_nearest(const int32_t* __restrict__ row, const int32_t* __restrict__ idx, const long long* __restrict__ dist,
                                     const int32_t* __restrict__ nf, int64_t n, int k, int64_t n_out, int32_t* __restrict__ o_idx,
                                     long long* __restrict__ o_dist, int32_t* __restrict__ o_nf, unsigned int* __restrict__ bad) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * k) return;
    const int64_t i = t / k;
    const int j = (int)(t - i * k);
    const int64_t r = row ? (int64_t)row[i] : i;
    if ((uint64_t)r >= (uint64_t)n_out) { atomicOr(bad, 1u); return; }
    o_idx[r * k + j] = idx[t];
    o_dist[r * k + j] = dist[t];
    if (j == 0) o_nf[r] = nf[i];
}

int pp_ensure(ivj_comm* c, size_t bytes) {
    if (bytes <= c->pp_cap) return IVJ_OK;
    if (c->pp_buf) (void)hipFree(c->pp_buf);
    c->pp_buf = nullptr; c->pp_cap = 0;
    const size_t want = align_up(bytes + bytes / 8, 1 << 20);
    if (hipMalloc((void**)&c->pp_buf, want) != hipSuccess) return fail(IVJ_ENOMEM, "scratch of the per-probe exchange: hipMalloc of " + std::to_string(want) + " bytes failed");
    c->pp_cap = want;
    return IVJ_OK;
}

// op: IVJ_STREAM_COUNT or IVJ_STREAM_NEAREST.  Failure protocol as overlap_allgather: every rank reaches the ONE count all-gather with
// {rows it reports | -1 = its own work failed, the n_total it was given}; whether anything is moved is decided from the gathered values
// alone (a failed rank: nobody sends or receives, the failed rank returns its own error, the others IVJ_EPEER; n_total differing between
// the ranks or fewer output rows than reported rows: IVJ_EINVAL on every rank), so no rank is ever left waiting in a collective.  Everything
// of this rank's own that can fail (the shard's kernel, allocations, the stream synchronisation, the flag words) happens BEFORE the count
// all-gather and is folded into the failure mark; between the all-gather and the grouped send / receive batch there is no fallible local
// step, and what follows the batch (merge kernels, flag read-back) no peer waits for.
int per_probe_allgather(ivj_comm* c, ivj_index* ix, const ivj_side* probe, const ivj_opts* opts, int op, int64_t n_total,
                        int64_t* counts_out, int32_t* idx_out, int64_t* dist_out, int32_t* nf_out) {
    ivj_ctx* ctx = c->ctx;
    const int64_t n = probe->n;
    const int k = op == IVJ_STREAM_NEAREST ? (opts->nearest_k < 1 ? 1 : opts->nearest_k) : 1;
    int rc = IVJ_OK;
    std::string main_err;
    auto note = [&](int r) { if (r != IVJ_OK && rc == IVJ_OK) { rc = r; main_err = g_err; } };
    if (c->world > 1 && n > 0 && !probe->row_id) note(fail(IVJ_EINVAL, "a shard of a multi-rank call needs the global probe rows in probe.row_id"));
    if (n > n_total) note(fail(IVJ_EINVAL, "the shard has more probe rows than n_total"));
    if (n_total > 0x7fffffffll) note(fail(IVJ_EINVAL, "n_total beyond int32 row ids"));
    if (rc == IVJ_OK && fault_injected(c->rank, 0)) note(fail(IVJ_EHIP, "injected fault (IVJ_FAULT_ALLGATHER) in the per-probe exchange"));
    const bool self_rccl = c->self_rccl && c->comm;
    const bool own_in_recv = self_rccl;                                   // this rank's slice travels through RCCL like everybody else's
    const bool need_recv = c->world > 1 || self_rccl;
    const bool make_rows = !probe->row_id && self_rccl;                  // the wire needs a row column; a shard without ids has the identity
    const int64_t ntiles = (n_total + PP_TILE - 1) / PP_TILE;
    // scratch layout: [flag words | tile offsets | send columns | receive columns (n_total rows; only where something is received)]
    const size_t n1 = (size_t)std::max<int64_t>(n, 1), nt1 = (size_t)std::max<int64_t>(n_total, 1);
    const size_t toff_b = align_up((size_t)std::min<int>(c->world, PP_MAX_WORLD) * (size_t)(ntiles + 1) * 4);
    const size_t a_row = align_up(n1 * 4), a_row_t = align_up(nt1 * 4);
    const size_t a_idx = align_up(n1 * k * 4), a_dist = align_up(n1 * k * 8), a_idx_t = align_up(nt1 * k * 4), a_dist_t = align_up(nt1 * k * 8);
    const size_t send_b = (make_rows ? a_row : 0) + (op == IVJ_STREAM_COUNT ? a_row : a_idx + a_dist + a_row);
    const size_t recv_b = !need_recv ? 0 : (op == IVJ_STREAM_COUNT ? 2 * a_row_t : a_row_t + a_idx_t + a_dist_t + a_row_t);
    if (rc == IVJ_OK) note(pp_ensure(c, 256 + toff_b + send_b + recv_b));
    char* base = c->pp_buf;
    unsigned int* d_flags = reinterpret_cast<unsigned int*>(base);
    uint32_t* d_toff = base ? reinterpret_cast<uint32_t*>(base + 256) : nullptr;
    char* p_send = base ? base + 256 + toff_b : nullptr;
    char* p_recv = p_send ? p_send + send_b : nullptr;
    const void* send[4] = {nullptr, nullptr, nullptr, nullptr};
    void* recv[4] = {nullptr, nullptr, nullptr, nullptr};
    int widths[4] = {4, 4, 8, 4};
    int n_cols = 2;
    const int32_t* own_rows = probe->row_id;
    if (rc == IVJ_OK) {
        char* q = p_send;
        if (make_rows) { own_rows = (int32_t*)q; q += a_row; }
        char* r = p_recv;
        send[0] = own_rows; recv[0] = r; r += a_row_t;
        if (op == IVJ_STREAM_COUNT) {
            int32_t* s_cnt = (int32_t*)q;
            send[1] = s_cnt; recv[1] = r;
            widths[0] = 4; widths[1] = 4; n_cols = 2;
            if (n > 0) note(count_overlaps_dev(ctx, ix, probe, opts, nullptr, s_cnt));     // the wire column straight from the kernel
        } else {
            int32_t* s_idx = (int32_t*)q; q += a_idx;
            int64_t* s_dist = (int64_t*)q; q += a_dist;
            int32_t* s_nf = (int32_t*)q;
            recv[1] = r; r += a_idx_t;
            recv[2] = r; r += a_dist_t;
            recv[3] = r;
            send[1] = s_idx; send[2] = s_dist; send[3] = s_nf;
            widths[0] = 4; widths[1] = 4 * k; widths[2] = 8 * k; widths[3] = 4; n_cols = 4;
            if (n > 0) note(nearest_dev(ctx, ix, probe, opts, s_idx, s_dist, s_nf));
        }
        if (rc == IVJ_OK && make_rows && n > 0) {
            hipLaunchKernelGGL(k_pp_rows, dim3(grid1d(n, 256)), dim3(256), 0, ctx->stream, (const int32_t*)nullptr, n, (int32_t*)own_rows);
            if (hipGetLastError() != hipSuccess) note(fail(IVJ_EHIP, "k_pp_rows launch failed"));
        }
        if (rc == IVJ_OK && hipMemsetAsync(d_flags, 0, 16, ctx->stream) != hipSuccess) note(fail(IVJ_EHIP, "the flag words of the per-probe exchange: hipMemsetAsync failed"));
        // (also covers whatever the context's stream still holds for the caller's output columns)
        if (rc == IVJ_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) note(fail(IVJ_EHIP, "the shard's per-probe kernel failed"));
    }
    // every rank, whatever happened to it so far
    std::vector<int64_t> all((size_t)c->world * 2), counts((size_t)c->world);
    const int64_t mine[2] = {rc == IVJ_OK ? n : -1, n_total};
    IVJ_TRY(comm_allgather_i64(c, mine, 2, all.data()));
    int failed_peer = -1;
    bool mismatch = false;
    int64_t tot = 0;
    for (int r = 0; r < c->world; ++r) {
        const int64_t nr = all[(size_t)r * 2];
        if (nr < 0 && failed_peer < 0) failed_peer = r;
        if (all[(size_t)r * 2 + 1] != n_total) mismatch = true;
        counts[r] = nr < 0 ? 0 : nr;
        tot += counts[r];
    }
    if (rc != IVJ_OK) { g_err = main_err; return rc; }
    if (failed_peer >= 0) return fail(IVJ_EPEER, "rank " + std::to_string(failed_peer) + " failed in the per-probe exchange; nothing was exchanged");
    if (mismatch) return fail(IVJ_EINVAL, "the ranks of a per-probe exchange disagree on n_total");
    if (tot > n_total) return fail(IVJ_EINVAL, "the ranks report " + std::to_string(tot) + " probe rows for n_total = " + std::to_string(n_total));
    // the ONE grouped batch: the peers' slices into the receive columns (this rank's own stays where the kernel wrote it)
    if (need_recv) IVJ_TRY(comm_exchange_v(c, send, recv, n_cols, widths, counts.data(), 0, own_in_recv));
    if (n_total == 0) { if (need_recv) HIP_TRY(hipStreamSynchronize(c->xstream)); return IVJ_OK; }
    std::vector<int64_t> off((size_t)c->world + 1, 0);
    for (int r = 0; r < c->world; ++r) off[r + 1] = off[r] + counts[r];
    auto read_flags = [&](unsigned int* f) -> int {
        HIP_TRY(hipMemcpyAsync(c->h_counts, d_flags, 16, hipMemcpyDeviceToHost, c->xstream));
        HIP_TRY(hipStreamSynchronize(c->xstream));
        std::memcpy(f, c->h_counts, 16);
        return IVJ_OK;
    };
    const int32_t* r_rows = (const int32_t*)recv[0];
    bool merged = false;
    if (c->world <= PP_MAX_WORLD) {
        PpSrc S;
        S.world = c->world; S.own = own_in_recv ? -1 : c->rank;
        for (int r = 0; r <= c->world; ++r) S.off[r] = off[r];
        hipLaunchKernelGGL(k_pp_tile_offsets, dim3(grid1d((int64_t)c->world * (ntiles + 1), 256)), dim3(256), 0, c->xstream, S, r_rows, own_rows, ntiles, d_toff);
        if (op == IVJ_STREAM_COUNT)
            hipLaunchKernelGGL(k_pp_merge_count, dim3((unsigned)ntiles), dim3(PP_THREADS), 0, c->xstream, S, r_rows, own_rows, (const int32_t*)recv[1], (const int32_t*)send[1],
                               n_total, ntiles, (const uint32_t*)d_toff, (long long*)counts_out, d_flags);
        else
            hipLaunchKernelGGL(k_pp_merge_nearest, dim3((unsigned)ntiles), dim3(PP_THREADS), 0, c->xstream, S, r_rows, own_rows, (const int32_t*)recv[1], (const int32_t*)send[1],
                               (const long long*)recv[2], (const long long*)send[2], (const int32_t*)recv[3], (const int32_t*)send[3], k, n_total, ntiles,
                               (const uint32_t*)d_toff, idx_out, (long long*)dist_out, nf_out, d_flags);
        HIP_TRY(hipGetLastError());
        unsigned int f[4];
        IVJ_TRY(read_flags(f));
        unsigned long long placed; std::memcpy(&placed, f + 2, 8);
        if (f[0] & 2u) return fail(IVJ_EINVAL, "a probe row id is reported twice in the per-probe exchange (by one rank or by two)");
        merged = f[0] == 0 && placed == (unsigned long long)tot;
    }
    if (merged) return IVJ_OK;
    // senders that are not ascending (or more ranks than the merge kernels take): defaults, then one store per reported row
    ++c->pp_scatter_fallbacks;
    HIP_TRY(hipMemsetAsync(d_flags, 0, 16, c->xstream));
    if (op == IVJ_STREAM_COUNT) HIP_TRY(hipMemsetAsync(counts_out, 0, (size_t)n_total * 8, c->xstream));
    else {
        HIP_TRY(hipMemsetAsync(idx_out, 0xff, (size_t)n_total * k * 4, c->xstream));       // -1
        HIP_TRY(hipMemsetAsync(dist_out, 0xff, (size_t)n_total * k * 8, c->xstream));      // -1
        HIP_TRY(hipMemsetAsync(nf_out, 0, (size_t)n_total * 4, c->xstream));
    }
    auto scatter = [&](const int32_t* rows, const void* c1, const void* c2, const void* c3, int64_t m) {
        if (m <= 0) return;
        if (op == IVJ_STREAM_COUNT)
            hipLaunchKernelGGL(k_pp_scatter_count, dim3(grid1d(m, 256)), dim3(256), 0, c->xstream, rows, (const int32_t*)c1, m, n_total, (long long*)counts_out, d_flags);
        else
            hipLaunchKernelGGL(k_pp_scatter_nearest, dim3(grid1d(m * k, 256)), dim3(256), 0, c->xstream, rows, (const int32_t*)c1, (const long long*)c2, (const int32_t*)c3,
                               m, k, n_total, idx_out, (long long*)dist_out, nf_out, d_flags);
    };
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank && !own_in_recv) { scatter(own_rows, send[1], send[2], send[3], counts[r]); continue; }
        scatter(r_rows + off[r], (const char*)recv[1] + (size_t)off[r] * widths[1], op == IVJ_STREAM_COUNT ? nullptr : (const char*)recv[2] + (size_t)off[r] * widths[2],
                op == IVJ_STREAM_COUNT ? nullptr : (const char*)recv[3] + (size_t)off[r] * widths[3], counts[r]);
    }
    HIP_TRY(hipGetLastError());
    unsigned int f[4];
    IVJ_TRY(read_flags(f));
    if (f[0]) return fail(IVJ_EINVAL, "a probe row id of the per-probe exchange lies outside [0, n_total)");
    return IVJ_OK;
}

}  // namespace
