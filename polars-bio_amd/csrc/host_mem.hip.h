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
        const unsigned nt = hw >= 16 ? 16u : (hw ? hw : 1u);
        std::vector<std::thread> th;
        const size_t per = (sz / nt + 4095) / 4096 * 4096;
        for (unsigned t = 0; t < nt; ++t) {
            const size_t lo = (size_t)t * per, hi = lo + per < sz ? lo + per : sz;
            if (lo >= hi) break;
            th.emplace_back([p, lo, hi] { volatile char* c = (volatile char*)p; for (size_t o = lo; o < hi; o += 4096) c[o] = 0; });
        }
        for (auto& x : th) x.join();
    }
    return p;
}

// first touch of a caller-provided result range by a few threads (a DMA write into never-touched pages faults page by page)
void host_prefault(void* p, size_t bytes) {
    if (!p || bytes < ((size_t)8 << 20)) return;
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned nt = hw >= 16 ? 16u : (hw ? hw : 1u);
    std::vector<std::thread> th;
    const size_t per = (bytes / nt + 4095) / 4096 * 4096;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t lo = (size_t)t * per, hi = lo + per < bytes ? lo + per : bytes;
        if (lo >= hi) break;
        th.emplace_back([p, lo, hi] { volatile char* c = (volatile char*)p; for (size_t o = lo; o < hi; o += 4096) c[o] = c[o]; c[hi - 1] = c[hi - 1]; });
    }
    for (auto& x : th) x.join();
}

// copy into a pinned staging slot with a few threads (one thread moves ~31 GB/s, four ~95 GB/s: tools/pcie_probe.py)
void host_copy_parallel(void* dst, const void* src, size_t bytes) {
    if (bytes < ((size_t)8 << 20)) { std::memcpy(dst, src, bytes); return; }
    constexpr unsigned nt = 4;
    std::thread th[nt - 1];
    const size_t per = (bytes / nt + 4095) / 4096 * 4096;
    for (unsigned t = 1; t < nt; ++t) {
        const size_t lo = (size_t)t * per, hi = lo + per < bytes ? lo + per : bytes;
        th[t - 1] = std::thread([=] { if (lo < hi) std::memcpy((char*)dst + lo, (const char*)src + lo, hi - lo); });
    }
    std::memcpy(dst, src, per < bytes ? per : bytes);
    for (auto& x : th) x.join();
}

// registers a host range for the lifetime of the object (failure is not an error: the copy then goes the pageable way)
struct HostPin {
    void* p = nullptr;
    bool ok = false;
    hipError_t code = hipSuccess;
    HostPin(const void* ptr, size_t bytes, size_t min_bytes = (size_t)1 << 20) {
        if (ptr && bytes >= min_bytes) {
            p = const_cast<void*>(ptr);
            code = hipHostRegister(p, bytes, hipHostRegisterDefault);
            ok = code == hipSuccess;
            if (!ok) (void)hipGetLastError();
        }
    }
    ~HostPin() { if (ok) (void)hipHostUnregister(p); }
    HostPin(const HostPin&) = delete;
    HostPin& operator=(const HostPin&) = delete;
};

// Every copy between caller / library host memory and HBM goes through this object, never through the runtime's own handling
// of pageable memory: left to itself the runtime pins a pageable range on the fly and keeps up to eight such pins per queue
// cached BY ADDRESS -- a later copy whose host range starts where an earlier, since freed one did reuses the stale pin (an H2D
// source is pinned read-only: "Memory access fault ... Write access to a read-only page" when malloc hands the same address to a
// result buffer; pages that were unmapped in between: "Reason: Unknown").  Here a range of XFER_PIN_MIN bytes or more is
// registered for the duration of the copy (hipHostRegister, ~3 ms / GB; a range the caller has registered already is used as it
// is), anything smaller -- or not registrable -- is staged through the context's own pinned bounce buffer.
// add copies, then finish(): when it returns every copy is complete and every range unregistered.
constexpr size_t XFER_PIN_MIN = (size_t)256 << 10;
constexpr size_t XFER_BOUNCE = (size_t)4 << 20;

struct HostXfer {
    hipStream_t stream;
    char** bounce;                                         // the context's pinned bounce buffer (allocated on first use)
    std::vector<HostPin*> pins;
    struct Pend { void* dst; size_t off, bytes; };
    std::vector<Pend> outs;                                // D2H copies parked in the bounce buffer
    size_t used = 0;
    hipError_t err = hipSuccess;
    HostXfer(hipStream_t s, char** bounce_slot) : stream(s), bounce(bounce_slot) {}
    HostXfer(const HostXfer&) = delete;
    HostXfer& operator=(const HostXfer&) = delete;
    ~HostXfer() { (void)finish(); }

    bool direct(const void* host, size_t bytes) {          // true: the DMA engine may address the range itself
        if (bytes < XFER_PIN_MIN) return false;
        HostPin* p = new HostPin(host, bytes, XFER_PIN_MIN);
        if (p->ok) { pins.push_back(p); return true; }
        const bool already = p->code == hipErrorHostMemoryAlreadyRegistered;
        delete p;
        return already;
    }
    bool have_bounce() {
        if (!*bounce && err == hipSuccess) err = hipHostMalloc((void**)bounce, XFER_BOUNCE, hipHostMallocDefault);
        return *bounce != nullptr;
    }
    void drain() {                                         // completes what is parked in the bounce buffer
        const hipError_t s = hipStreamSynchronize(stream);
        if (err == hipSuccess) err = s;
        if (err == hipSuccess) for (const Pend& o : outs) std::memcpy(o.dst, *bounce + o.off, o.bytes);
        outs.clear();
        used = 0;
    }
    void h2d(void* dst_dev, const void* src_host, size_t bytes) {
        if (err != hipSuccess || bytes == 0) return;
        if (direct(src_host, bytes)) { err = hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, stream); return; }
        if (!have_bounce()) return;
        for (size_t o = 0; o < bytes && err == hipSuccess;) {
            if (used == XFER_BOUNCE) drain();
            const size_t n = std::min(bytes - o, XFER_BOUNCE - used);
            std::memcpy(*bounce + used, (const char*)src_host + o, n);
            err = hipMemcpyAsync((char*)dst_dev + o, *bounce + used, n, hipMemcpyHostToDevice, stream);
            used += (n + 255) & ~(size_t)255; if (used > XFER_BOUNCE) used = XFER_BOUNCE;
            o += n;
        }
    }
    void d2h(void* dst_host, const void* src_dev, size_t bytes) {
        if (err != hipSuccess || bytes == 0) return;
        if (direct(dst_host, bytes)) { err = hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, stream); return; }
        if (!have_bounce()) return;
        for (size_t o = 0; o < bytes && err == hipSuccess;) {
            if (used == XFER_BOUNCE) drain();
            const size_t n = std::min(bytes - o, XFER_BOUNCE - used);
            err = hipMemcpyAsync(*bounce + used, (const char*)src_dev + o, n, hipMemcpyDeviceToHost, stream);
            outs.push_back(Pend{(char*)dst_host + o, used, n});
            used += (n + 255) & ~(size_t)255; if (used > XFER_BOUNCE) used = XFER_BOUNCE;
            o += n;
        }
    }
    hipError_t finish() {
        if (used || !pins.empty() || !outs.empty()) drain();
        for (HostPin* p : pins) delete p;
        pins.clear();
        return err;
    }
};

}  // namespace

