// host_mem.hip.h -- host <-> HBM traffic of the host-buffer entry points: registered inputs, pre-faulted registered results
// Part of the single translation unit ivjoin.hip (included there, in this order); not a stand-alone header.
//
// Measured on the MI355X box (tools/pcie_probe.py, profiles/r02): pageable and registered host memory both move at
// ~57 GB/s over PCIe Gen5, hipHostRegister of 1.2 GB costs ~3 ms -- but a D2H copy into FRESH (never touched) host pages
// runs at 7 GB/s, because every 4-KiB page faults under the DMA.  The round-1 host path paid that twice (library malloc +
// numpy copy): 0.36 s for config 3, of which the join was 4 ms.  Here result buffers are 2-MiB aligned, advised to huge
// pages, first-touched by a few host threads and registered, and the caller's input columns are registered in place
// (zero copy: the DMA engine reads the Arrow / numpy buffers directly).
#pragma once

#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>

namespace {

// Host memory a new allocation can use right now (bytes); 0 when it cannot be told.  MemAvailable of /proc/meminfo counts the
// reclaimable page cache (a host that has just read large Parquet / CSV inputs has little MemFree but tens of GB available);
// sysconf(_SC_AVPHYS_PAGES) = MemFree is only the fallback.  IVJ_HOST_MEM_AVAILABLE (bytes) overrides both (tests).
size_t parse_meminfo_available(const char* text) {
    const char* p = text ? std::strstr(text, "MemAvailable:") : nullptr;
    if (!p) return 0;
    p += 13;
    while (*p == ' ' || *p == '\t') ++p;
    char* end = nullptr;
    const unsigned long long kb = std::strtoull(p, &end, 10);
    return end == p ? 0 : (size_t)kb * 1024;
}
size_t host_mem_available() {
    if (const char* o = std::getenv("IVJ_HOST_MEM_AVAILABLE")) { const unsigned long long v = std::strtoull(o, nullptr, 10); if (v) return (size_t)v; }
    if (FILE* f = std::fopen("/proc/meminfo", "r")) {
        char buf[2048];
        const size_t n = std::fread(buf, 1, sizeof(buf) - 1, f);
        std::fclose(f);
        buf[n] = 0;
        const size_t a = parse_meminfo_available(buf);
        if (a) return a;
    }
    const long pages = sysconf(_SC_AVPHYS_PAGES), psz = sysconf(_SC_PAGESIZE);
    return (pages > 0 && psz > 0) ? (size_t)pages * (size_t)psz : 0;
}

// ---- host worker pool ---------------------------------------------------------------------------------------------------------------
// The host passes of this library (staging copies of HostXfer, first touch of result buffers, the front door's per-row passes) are
// short: 0.1 - 3 ms of work cut into a few dozen parts, hundreds of times per call.  Spawning std::threads for each (~25 us apiece,
// one after the other) cost config 3's host path ~13 ms of 65 and pb.overlap ~8 ms of 40.  host_parallel(parts, f) runs f(0 ..
// parts - 1) on a process-wide pool of workers that sleep on a condition variable between jobs; the caller takes parts too.  One job
// at a time: a second caller that finds the pool busy (the front door calls in from several Python threads) runs its parts on
// threads of its own, as before.  A forked child starts with a fresh pool (the parent's workers do not exist in it).
struct HostPool {
    std::mutex run;                                        // one job at a time
    std::mutex m;                                          // guards everything below
    std::condition_variable cv_job, cv_done;
    int n_workers = 0;
    const std::function<void(int)>* job = nullptr;
    unsigned long long gen = 0;
    int parts = 0;
    std::atomic<int> next{0};
    int pending = 0;                                       // parts not yet finished
    int active = 0;                                        // workers that picked this job up and have not let go of it yet
    std::exception_ptr error;                              // the first exception a part threw (rethrown by execute() on the caller's thread)
    static constexpr int MAX_WORKERS = 63;

    // grab parts until none is left; `worker`: the caller of execute() is not counted in `active`.  A part that throws (std::bad_alloc
    // out of a std::vector growing inside a front-door pass) must neither terminate the process from a detached worker nor leave
    // `pending` unsettled: the exception is parked, the part counts as done, and the parts still unclaimed are skipped.
    void work(const std::function<void(int)>& f, int n, bool worker) {
        int done = 0;
        std::exception_ptr mine;
        for (int i = next.fetch_add(1, std::memory_order_relaxed); i < n; i = next.fetch_add(1, std::memory_order_relaxed)) {
            if (!mine) {
                try { f(i); } catch (...) { mine = std::current_exception(); }
            }
            ++done;
        }
        std::lock_guard<std::mutex> g(m);
        if (mine && !error) error = mine;
        pending -= done;
        if (worker) --active;
        if (pending == 0 && active == 0) cv_done.notify_all();
    }
    void worker_main() {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(int)>* f;
            int n;
            {
                std::unique_lock<std::mutex> g(m);
                cv_job.wait(g, [&] { return gen != seen; });
                seen = gen; f = job; n = parts;
                if (f) ++active;                           // execute() does not return (and `f` stays alive) until we let go
            }
            if (f) work(*f, n, true);
        }
    }
    void ensure_workers(int want) {                        // called with `run` held
        if (want > MAX_WORKERS) want = MAX_WORKERS;
        while (n_workers < want) {
            std::thread([this] { worker_main(); }).detach();   // process-lifetime workers: never joined (no static-destructor order games)
            ++n_workers;
        }
    }
    void execute(int n, const std::function<void(int)>& f) {
        ensure_workers(n - 1);
        {
            std::lock_guard<std::mutex> g(m);
            job = &f; parts = n; pending = n; next.store(0, std::memory_order_relaxed); ++gen;
        }
        // wake as many workers as there are parts besides the caller's (not all 63: a four-part staging copy must not pay for a
        // stampede on `m`); a worker that stays asleep picks up whatever job is current when it is next woken
        for (int k = 0; k < n - 1 && k < n_workers; ++k) cv_job.notify_one();
        work(f, n, false);
        std::exception_ptr err;
        {
            std::unique_lock<std::mutex> g(m);
            cv_done.wait(g, [&] { return pending == 0 && active == 0; });
            job = nullptr;                                 // a worker that wakes up late finds no job
            err = error; error = nullptr;
        }
        if (err) std::rethrow_exception(err);              // on the caller's thread, where IVJ_HOST_GUARD / IVJ_ABI_CATCH turn it into a status
    }
};

inline HostPool*& host_pool_slot() { static HostPool* p = nullptr; return p; }
inline std::mutex*& host_pool_init_mutex() { static std::mutex* m = new std::mutex(); return m; }
// the child must not touch the parent's pool (its threads are gone) NOR the init mutex, which another thread of the parent may have
// held at the fork: the child gets a fresh one (both old objects are leaked, a few bytes per fork)
inline void host_pool_after_fork() { host_pool_slot() = nullptr; host_pool_init_mutex() = new std::mutex(); }
inline HostPool* host_pool() {
    std::lock_guard<std::mutex> g(*host_pool_init_mutex());
    HostPool*& p = host_pool_slot();
    if (!p) {
        static bool hooked = false;
        if (!hooked) { (void)pthread_atfork(nullptr, nullptr, host_pool_after_fork); hooked = true; }
        p = new HostPool();                                // leaked on purpose: lives as long as the process
    }
    return p;
}

// f(0 .. parts - 1), in parallel; returns when all have run
inline void host_parallel(int parts, const std::function<void(int)>& f) {
    if (parts <= 1) { if (parts == 1) f(0); return; }
    static thread_local bool inside = false;               // a part that calls host_parallel itself must not touch the pool's lock again
    if (!inside) {
        HostPool* pool = host_pool();
        std::unique_lock<std::mutex> busy(pool->run, std::try_to_lock);
        if (busy.owns_lock()) {
            inside = true;
            try { pool->execute(parts, f); } catch (...) { inside = false; throw; }
            inside = false;
            return;
        }
    }
    std::vector<std::thread> th;                           // pool busy with another caller's job: threads of our own
    th.reserve(parts - 1);
    std::mutex em;
    std::exception_ptr err;                                // an exception inside a std::thread body would be std::terminate
    auto guarded = [&](int k) { try { f(k); } catch (...) { std::lock_guard<std::mutex> g(em); if (!err) err = std::current_exception(); } };
    for (int k = 1; k < parts; ++k) th.emplace_back([&guarded, k] { guarded(k); });
    guarded(0);
    for (auto& x : th) x.join();
    if (err) std::rethrow_exception(err);
}

// A result that cannot fit the host is refused with an error instead of being first-touched into the OOM killer (an
// all-against-all join of 400 k x 600 k rows is 1.7e10 pairs = 139 GB: that took two GPU boxes down in round 2).
bool host_result_fits(size_t bytes) {
    const size_t avail = host_mem_available();
    return avail == 0 || bytes <= avail - avail / 8;
}

// host memory for a result column: huge-page friendly, pre-faulted in parallel.  Free with std::free.
void* host_result_alloc(size_t bytes) {
    if (bytes == 0) bytes = 1;
    const size_t huge = (size_t)2 << 20;
    const size_t sz = bytes >= huge ? (bytes + huge - 1) / huge * huge : (bytes + 4095) / 4096 * 4096;
    void* p = nullptr;
    if (posix_memalign(&p, bytes >= huge ? huge : 4096, sz) != 0) return nullptr;
    if (bytes >= huge) (void)madvise(p, sz, MADV_HUGEPAGE);
    if (bytes >= ((size_t)8 << 20)) {
        unsigned hw = std::thread::hardware_concurrency();
        const int nt = hw >= 16 ? 16 : (hw ? (int)hw : 1);
        const size_t per = (sz / nt + 4095) / 4096 * 4096;
        host_parallel(nt, [p, per, sz](int t) {
            const size_t lo = (size_t)t * per, hi = lo + per < sz ? lo + per : sz;
            volatile char* c = (volatile char*)p;
            for (size_t o = lo; o < hi; o += 4096) c[o] = 0;
        });
    }
    return p;
}

// first touch of a caller-provided result range by a few threads (a DMA write into never-touched pages faults page by page)
void host_prefault(void* p, size_t bytes) {
    if (!p || bytes < ((size_t)8 << 20)) return;
    unsigned hw = std::thread::hardware_concurrency();
    const int nt = hw >= 16 ? 16 : (hw ? (int)hw : 1);
    const size_t per = (bytes / nt + 4095) / 4096 * 4096;
    host_parallel(nt, [p, per, bytes](int t) {
        const size_t lo = (size_t)t * per, hi = lo + per < bytes ? lo + per : bytes;
        if (lo >= hi) return;
        volatile char* c = (volatile char*)p;
        for (size_t o = lo; o < hi; o += 4096) c[o] = c[o];
        c[hi - 1] = c[hi - 1];
    });
}

// copy into a pinned staging slot with a few threads (one thread moves ~31 GB/s, four ~95 GB/s: tools/pcie_probe.py)
void host_copy_parallel(void* dst, const void* src, size_t bytes) {
    if (bytes < ((size_t)8 << 20)) { std::memcpy(dst, src, bytes); return; }
    constexpr int nt = 4;
    const size_t per = (bytes / nt + 4095) / 4096 * 4096;
    host_parallel(nt, [=](int t) {
        const size_t lo = (size_t)t * per, hi = lo + per < bytes ? lo + per : bytes;
        if (lo < hi) std::memcpy((char*)dst + lo, (const char*)src + lo, hi - lo);
    });
}

// Every copy between caller / library host memory and HBM goes through this object and the context's own PINNED staging
// slots -- never through a host pointer the runtime has to pin, and never through hipHostRegister.  Left to itself the runtime pins
// a pageable range on the fly and keeps such pins cached BY ADDRESS (eight per queue, in one process-wide map); torch does the same
// for its pageable copies.  A later copy -- or a later hipHostRegister, which then reports "already registered" -- whose range
// overlaps an address a since-freed buffer used to occupy runs through the stale mapping: "Memory access fault ... Write access to a
// read-only page" (the stale pin was an H2D source) or "Reason: Unknown" (the pages are gone), both seen in the GPU test suite once
// torch and this library had shared a process for a few hundred calls.  So: two slots of XFER_SLOT bytes; an H2D chunk is copied
// into a slot by a few host threads and sent from there, a D2H chunk lands in a slot and is copied out while the next chunk's DMA
// runs; small copies share a slot.  The DMA moves ~57 GB/s, four copy threads ~95 GB/s (tools/pcie_probe.py): the pipeline stays
// DMA-bound.  add copies, then finish(): when it returns every copy is complete.
constexpr size_t XFER_SLOT = (size_t)16 << 20;

struct HostXfer {
    hipStream_t stream;
    XferSlots* slots;
    struct Pend { void* dst; size_t off, bytes; };
    std::vector<Pend> outs[2];                             // D2H chunks parked in a slot, not yet copied out
    bool busy[2] = {false, false};                         // the slot has DMA in flight behind its event
    int cur = 0;
    size_t used = 0;
    hipError_t err = hipSuccess;
    HostXfer(hipStream_t s, XferSlots* sl) : stream(s), slots(sl) {}
    HostXfer(const HostXfer&) = delete;
    HostXfer& operator=(const HostXfer&) = delete;
    ~HostXfer() { (void)finish(); }

    bool ready() {
        for (int k = 0; k < 2 && err == hipSuccess; ++k) {
            if (!slots->buf[k]) err = hipHostMalloc((void**)&slots->buf[k], XFER_SLOT, hipHostMallocDefault);
            if (err == hipSuccess && !slots->ev[k]) err = hipEventCreateWithFlags(&slots->ev[k], hipEventDisableTiming);
        }
        return err == hipSuccess;
    }
    void settle(int k) {                                   // slot k: its DMA is done, its D2H chunks are in the caller's memory
        if (busy[k]) {
            const hipError_t e = hipEventSynchronize(slots->ev[k]);
            if (err == hipSuccess) err = e;
            busy[k] = false;
        }
        if (err == hipSuccess) for (const Pend& o : outs[k]) host_copy_parallel(o.dst, slots->buf[k] + o.off, o.bytes);
        outs[k].clear();
    }
    void rotate() {                                        // the current slot is full: mark it, take the other one
        if (err == hipSuccess) err = hipEventRecord(slots->ev[cur], stream);
        busy[cur] = true;
        cur ^= 1;
        settle(cur);
        used = 0;
    }
    size_t reserve(size_t want, size_t& off) {             // a piece of the current slot (256-byte granules)
        if (used >= XFER_SLOT) rotate();
        off = used;
        const size_t n = want < XFER_SLOT - used ? want : XFER_SLOT - used;
        used += (n + 255) & ~(size_t)255;
        return n;
    }
    void h2d(void* dst_dev, const void* src_host, size_t bytes) {
        if (err != hipSuccess || bytes == 0 || !ready()) return;
        for (size_t o = 0; o < bytes && err == hipSuccess;) {
            size_t off;
            const size_t n = reserve(bytes - o, off);
            host_copy_parallel(slots->buf[cur] + off, (const char*)src_host + o, n);
            err = hipMemcpyAsync((char*)dst_dev + o, slots->buf[cur] + off, n, hipMemcpyHostToDevice, stream);
            o += n;
        }
    }
    void d2h(void* dst_host, const void* src_dev, size_t bytes) {
        if (err != hipSuccess || bytes == 0 || !ready()) return;
        for (size_t o = 0; o < bytes && err == hipSuccess;) {
            size_t off;
            const size_t n = reserve(bytes - o, off);
            err = hipMemcpyAsync(slots->buf[cur] + off, (const char*)src_dev + o, n, hipMemcpyDeviceToHost, stream);
            outs[cur].push_back(Pend{(char*)dst_host + o, off, n});
            o += n;
        }
    }
    hipError_t finish() {
        if (used || busy[0] || busy[1] || !outs[0].empty() || !outs[1].empty()) {
            const hipError_t s = hipStreamSynchronize(stream);
            if (err == hipSuccess) err = s;
            busy[0] = busy[1] = false;
            settle(cur ^ 1);                               // the older slot first (row order of a result does not depend on it, cache warmth does)
            settle(cur);
            used = 0;
        }
        return err;
    }
};

}  // namespace

