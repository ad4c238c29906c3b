// host_frontdoor.hip.h -- host-side helpers of the front door: the per-row passes either side of the device path, in C++
// Part of the single translation unit ivjoin.hip (included there inside extern "C"); not a stand-alone header.
//
// What the reference does in its Rust executor around the join (DataFusion: dictionary / string handling of the join key,
// the column gathers of the renaming SELECT, src/operation.rs:272-301) the Python front door did with numpy / pyarrow.compute
// calls on a thread pool; three of those passes hold the interpreter lock or walk the data several times.  Here each is ONE pass
// over the rows on plain std::thread workers: no device work, no context.
//   ivj_host_narrow_i32   coordinate column (8 / 4 / 2 / 1-byte integers) -> int32 + its min / max   (the reference's int32 limit)
//   ivj_host_encode_utf8  Arrow string column (offsets + bytes) -> dictionary ids in first-occurrence order + one row per value
//   ivj_host_encode_keys64 column of 64-bit keys (the object pointers of a pandas object column) -> ids + one row per distinct key
//   ivj_host_remap_i32    dictionary indices -> ids of the shared dictionary through a small table, + which entries occur
//   ivj_host_take         fixed-width gather dst[i] = src[idx[i]] (the non-key columns of the joined rows)
//   ivj_host_widen_i32    int32 -> int64 (key columns materialised on the device back to the frame's dtype)

namespace {

int fd_threads(int64_t n, int want, int64_t min_rows_per_thread) {
    unsigned hw = std::thread::hardware_concurrency();
    int t = want > 0 ? want : 32;
    if (hw && t > (int)hw) t = (int)hw;
    const int64_t cap = n / (min_rows_per_thread > 0 ? min_rows_per_thread : 1);
    if ((int64_t)t > cap) t = (int)(cap < 1 ? 1 : cap);
    return t < 1 ? 1 : t;
}

// runs fn(part, lo, hi) over [0, n) cut into `t` contiguous ranges (multiples of 64 rows) on the host worker pool
template <class F>
void fd_parallel(int64_t n, int t, F&& fn) {
    if (t <= 1) { fn(0, (int64_t)0, n); return; }
    const int64_t per = ((n + t - 1) / t + 63) / 64 * 64;
    const int parts = (int)((n + per - 1) / per);
    host_parallel(parts, [&fn, per, n](int k) {
        const int64_t lo = (int64_t)k * per, hi = lo + per < n ? lo + per : n;
        if (lo < hi) fn(k, lo, hi);
    });
}

template <class T>
void fd_narrow_range(const T* s, int64_t lo, int64_t hi, int32_t* d, long long& mn, long long& mx) {
    T a = s[lo], b = s[lo];
    for (int64_t i = lo; i < hi; ++i) {
        const T v = s[i];
        a = v < a ? v : a; b = v > b ? v : b;
        d[i] = (int32_t)v;
    }
    // unsigned 64-bit values above the int64 range saturate: anything beyond int32 is refused by the caller anyway
    mn = (std::is_unsigned<T>::value && sizeof(T) == 8 && (unsigned long long)a > (unsigned long long)INT64_MAX) ? INT64_MAX : (long long)a;
    mx = (std::is_unsigned<T>::value && sizeof(T) == 8 && (unsigned long long)b > (unsigned long long)INT64_MAX) ? INT64_MAX : (long long)b;
}

struct FdStr {                                             // one dictionary entry of a worker
    unsigned long long w0;                                 // its first eight bytes (zero padded)
    int64_t row;                                           // a row that holds the value
    int32_t len;
};
// the first min(len, 8) bytes of a value as one word; `safe` = an eight-byte load at p stays inside the buffer
inline unsigned long long fd_word(const unsigned char* p, int64_t len, bool safe) {
    unsigned long long w = 0;
    if (safe) {
        std::memcpy(&w, p, 8);
        if (len < 8) w &= (1ull << (8 * len)) - 1ull;      // len 0 -> 0
        return w;
    }
    for (int64_t i = 0; i < len && i < 8; ++i) w |= (unsigned long long)p[i] << (8 * i);
    return w;
}
inline unsigned long long fd_mix(unsigned long long w, int64_t len) {
    unsigned long long h = (w ^ (0x9e3779b97f4a7c15ull * (unsigned long long)(len + 1))) * 0xff51afd7ed558ccdull;
    return h ^ (h >> 31);
}

constexpr int FD_MAX_VALUES = 4096;                        // distinct chrom values the native encoder handles (more: the caller's fallback)
constexpr int FD_SLOTS = 16384;                            // open addressing, power of two, <= 25 % full

// chrom names are a few bytes long: a value is identified by (length, first eight bytes) and, only beyond eight bytes, the rest
template <class Off>
struct FdEncoder {
    const Off* off;
    const unsigned char* data;
    int64_t data_end;                                      // bytes of `data` the offsets reach
    std::vector<int32_t> slot;                             // -1 or local id
    std::vector<FdStr> vals;
    bool overflow = false;
    unsigned long long last_w0 = 0;                        // the previous row's value (inputs sorted by chrom hit it every time)
    int64_t last_len = -1;
    int32_t last_id = 0;
    FdEncoder(const Off* o, const unsigned char* d, int64_t end) : off(o), data(d), data_end(end), slot(FD_SLOTS, -1) {}
    inline int32_t id_of(int64_t row) {
        const int64_t a = (int64_t)off[row], len = (int64_t)off[row + 1] - a;
        const unsigned char* p = data + a;
        const unsigned long long w0 = fd_word(p, len, a + 8 <= data_end);
        if (len <= 8 && len == last_len && w0 == last_w0) return last_id;
        unsigned long long h = fd_mix(w0, len);
        for (int64_t i = 8; i < len; i += 8) h = fd_mix(h ^ fd_word(p + i, len - i, a + i + 8 <= data_end), len);
        int32_t id = -1;
        for (unsigned s = (unsigned)h & (FD_SLOTS - 1);; s = (s + 1) & (FD_SLOTS - 1)) {
            const int32_t v = slot[s];
            if (v < 0) {
                if ((int)vals.size() >= FD_MAX_VALUES) { overflow = true; return 0; }
                id = slot[s] = (int32_t)vals.size();
                vals.push_back(FdStr{w0, row, (int32_t)len});
                break;
            }
            const FdStr& e = vals[v];
            if (e.len == len && e.w0 == w0 && (len <= 8 || std::memcmp(data + (int64_t)off[e.row] + 8, p + 8, (size_t)(len - 8)) == 0)) { id = v; break; }
        }
        if (len <= 8) { last_w0 = w0; last_len = len; last_id = id; }
        return id;
    }
};

template <class Off>
int fd_encode(const Off* off, const unsigned char* data, const uint8_t* validity, int64_t bit0, int64_t n, int32_t* ids, int64_t* dict_rows,
              int32_t dict_cap, int32_t* n_values, int threads) {
    const int t = fd_threads(n, threads, 1 << 14);
    std::vector<FdEncoder<Off>*> enc(t, nullptr);
    const int64_t data_end = (int64_t)off[n];
    for (int k = 0; k < t; ++k) enc[k] = new FdEncoder<Off>(off, data, data_end);
    auto valid = [&](int64_t i) { const int64_t b = bit0 + i; return !validity || ((validity[b >> 3] >> (b & 7)) & 1); };
    fd_parallel(n, t, [&](int k, int64_t lo, int64_t hi) {
        FdEncoder<Off>& e = *enc[k];
        for (int64_t i = lo; i < hi && !e.overflow; ++i) ids[i] = valid(i) ? e.id_of(i) : -1;
    });
    // merge the workers' dictionaries in worker order (= first occurrence order over the rows) and renumber
    bool overflow = false;
    for (int k = 0; k < t; ++k) overflow |= enc[k]->overflow;
    FdEncoder<Off> glob(off, data, data_end);
    std::vector<std::vector<int32_t>> remap(t);
    for (int k = 0; k < t && !overflow; ++k) {
        remap[k].resize(enc[k]->vals.size());
        for (size_t v = 0; v < enc[k]->vals.size() && !overflow; ++v) {
            remap[k][v] = glob.id_of(enc[k]->vals[v].row);
            overflow |= glob.overflow;
        }
    }
    int rc = IVJ_OK;
    if (overflow || (int64_t)glob.vals.size() > (int64_t)dict_cap) rc = IVJ_ECAPACITY;
    else {
        bool identity = true;                              // one worker, or every worker met the values in the same order
        for (int k = 0; k < t; ++k) for (size_t v = 0; v < remap[k].size(); ++v) identity &= remap[k][v] == (int32_t)v;
        if (!identity)
            fd_parallel(n, t, [&](int k, int64_t lo, int64_t hi) {
                const int32_t* r = remap[k].data();
                for (int64_t i = lo; i < hi; ++i) if (ids[i] >= 0) ids[i] = r[ids[i]];
            });
        for (size_t v = 0; v < glob.vals.size(); ++v) dict_rows[v] = glob.vals[v].row;
        *n_values = (int32_t)glob.vals.size();
    }
    for (int k = 0; k < t; ++k) delete enc[k];
    return rc;
}

// dictionary over 64-bit keys (PyObject pointers of an object column): open addressing per worker, merged in worker order
struct FdKeyDict {
    std::vector<int32_t> slot;
    std::vector<unsigned long long> keys;
    std::vector<int64_t> rows;
    bool overflow = false;
    unsigned long long last = 0;
    int32_t last_id = -1;
    FdKeyDict() : slot(FD_SLOTS, -1) {}
    inline int32_t id_of(unsigned long long k, int64_t row) {
        if (last_id >= 0 && k == last) return last_id;
        unsigned long long h = k * 0x9e3779b97f4a7c15ull;
        h ^= h >> 29;
        for (unsigned s = (unsigned)h & (FD_SLOTS - 1);; s = (s + 1) & (FD_SLOTS - 1)) {
            const int32_t v = slot[s];
            if (v < 0) {
                if ((int)keys.size() >= FD_MAX_VALUES) { overflow = true; return 0; }
                slot[s] = (int32_t)keys.size();
                keys.push_back(k); rows.push_back(row);
                last = k; last_id = slot[s];
                return last_id;
            }
            if (keys[v] == k) { last = k; last_id = v; return v; }
        }
    }
};

template <class I>
void fd_remap_range(const I* idx, int64_t lo, int64_t hi, const int32_t* remap, int64_t remap_len, int32_t* out, uint8_t* seen, bool& bad) {
    for (int64_t i = lo; i < hi; ++i) {
        const long long v = (long long)idx[i];
        if (v < 0) { out[i] = -1; continue; }
        if (v >= remap_len) { bad = true; out[i] = -1; continue; }
        out[i] = remap[v];
        seen[v] = 1;
    }
}

template <class T>
void fd_take_range(const T* src, int64_t n_src, const int32_t* idx, int64_t lo, int64_t hi, T* dst) {
    constexpr int AHEAD = 16;
    for (int64_t i = lo; i < hi; ++i) {
        if (i + AHEAD < hi) { const int32_t j = idx[i + AHEAD]; if (j >= 0) __builtin_prefetch(src + j, 0, 0); }
        const int32_t j = idx[i];
        dst[i] = (j >= 0 && (int64_t)j < n_src) ? src[j] : T(0);
    }
}

}  // namespace

extern "C" {

static int ivj_host_narrow_i32_impl(const void* src, int32_t src_bytes, int32_t is_unsigned, int64_t n, int32_t* dst, int64_t* out_min, int64_t* out_max,
                        int32_t threads) {
    if (n < 0 || (n > 0 && (!src || !dst)) || !out_min || !out_max) return fail(IVJ_EINVAL, "narrow: bad argument");
    if (src_bytes != 1 && src_bytes != 2 && src_bytes != 4 && src_bytes != 8) return fail(IVJ_EINVAL, "narrow: src_bytes must be 1, 2, 4 or 8");
    *out_min = 0; *out_max = 0;
    if (n == 0) return IVJ_OK;
    const int t = fd_threads(n, threads, 1 << 15);
    std::vector<long long> mn(t, INT64_MAX), mx(t, INT64_MIN);
    fd_parallel(n, t, [&](int k, int64_t lo, int64_t hi) {
        long long a = 0, b = 0;
        switch (src_bytes * 2 + (is_unsigned ? 1 : 0)) {
            case 16: fd_narrow_range((const int64_t*)src, lo, hi, dst, a, b); break;
            case 17: fd_narrow_range((const uint64_t*)src, lo, hi, dst, a, b); break;
            case 8: fd_narrow_range((const int32_t*)src, lo, hi, dst, a, b); break;
            case 9: fd_narrow_range((const uint32_t*)src, lo, hi, dst, a, b); break;
            case 4: fd_narrow_range((const int16_t*)src, lo, hi, dst, a, b); break;
            case 5: fd_narrow_range((const uint16_t*)src, lo, hi, dst, a, b); break;
            case 2: fd_narrow_range((const int8_t*)src, lo, hi, dst, a, b); break;
            default: fd_narrow_range((const uint8_t*)src, lo, hi, dst, a, b); break;
        }
        mn[k] = a; mx[k] = b;
    });
    long long a = INT64_MAX, b = INT64_MIN;
    for (int k = 0; k < t; ++k) { if (mn[k] <= mx[k]) { a = mn[k] < a ? mn[k] : a; b = mx[k] > b ? mx[k] : b; } }
    *out_min = a; *out_max = b;
    return IVJ_OK;
}

static int ivj_host_encode_utf8_impl(const void* offsets, int32_t offset_bytes, const uint8_t* data, const uint8_t* validity, int64_t validity_bit0, int64_t n,
                         int32_t* ids, int64_t* dict_rows, int32_t dict_cap, int32_t* n_values, int32_t threads) {
    if (n < 0 || !n_values || (n > 0 && (!offsets || !ids || !dict_rows))) return fail(IVJ_EINVAL, "encode: bad argument");
    if (offset_bytes != 4 && offset_bytes != 8) return fail(IVJ_EINVAL, "encode: offset_bytes must be 4 or 8");
    *n_values = 0;
    if (n == 0) return IVJ_OK;
    static const unsigned char none = 0;
    const unsigned char* d = data ? data : &none;          // a column of empty strings may come without a data buffer
    const int rc = offset_bytes == 4 ? fd_encode((const int32_t*)offsets, d, validity, validity_bit0, n, ids, dict_rows, dict_cap, n_values, threads)
                                     : fd_encode((const int64_t*)offsets, d, validity, validity_bit0, n, ids, dict_rows, dict_cap, n_values, threads);
    if (rc == IVJ_ECAPACITY) return fail(IVJ_ECAPACITY, "encode: more distinct values than the native encoder holds");
    return rc;
}

static int ivj_host_encode_keys64_impl(const uint64_t* keys, int64_t n, int32_t* ids, int64_t* dict_rows, int32_t dict_cap, int32_t* n_values, int32_t threads) {
    if (n < 0 || !n_values || (n > 0 && (!keys || !ids || !dict_rows))) return fail(IVJ_EINVAL, "encode keys: bad argument");
    *n_values = 0;
    if (n == 0) return IVJ_OK;
    const int t = fd_threads(n, threads, 1 << 16);
    std::vector<FdKeyDict> enc(t);
    fd_parallel(n, t, [&](int k, int64_t lo, int64_t hi) {
        FdKeyDict& e = enc[k];
        for (int64_t i = lo; i < hi && !e.overflow; ++i) ids[i] = e.id_of((unsigned long long)keys[i], i);
    });
    bool overflow = false;
    for (int k = 0; k < t; ++k) overflow |= enc[k].overflow;
    FdKeyDict glob;
    std::vector<std::vector<int32_t>> remap(t);
    for (int k = 0; k < t && !overflow; ++k) {
        remap[k].resize(enc[k].keys.size());
        for (size_t v = 0; v < enc[k].keys.size() && !overflow; ++v) {
            remap[k][v] = glob.id_of(enc[k].keys[v], enc[k].rows[v]);
            overflow |= glob.overflow;
        }
    }
    if (overflow || (int64_t)glob.keys.size() > (int64_t)dict_cap) return fail(IVJ_ECAPACITY, "encode keys: more distinct values than the native encoder holds");
    bool identity = true;
    for (int k = 0; k < t; ++k) for (size_t v = 0; v < remap[k].size(); ++v) identity &= remap[k][v] == (int32_t)v;
    if (!identity)
        fd_parallel(n, t, [&](int k, int64_t lo, int64_t hi) {
            const int32_t* r = remap[k].data();
            for (int64_t i = lo; i < hi; ++i) ids[i] = r[ids[i]];
        });
    for (size_t v = 0; v < glob.keys.size(); ++v) dict_rows[v] = glob.rows[v];
    *n_values = (int32_t)glob.keys.size();
    return IVJ_OK;
}

static int ivj_host_remap_i32_impl(const void* idx, int32_t idx_bytes, int64_t n, const int32_t* remap, int64_t remap_len, int32_t* out, uint8_t* seen,
                       int32_t threads) {
    if (n < 0 || remap_len < 0 || (n > 0 && (!idx || !out)) || (remap_len > 0 && (!remap || !seen))) return fail(IVJ_EINVAL, "remap: bad argument");
    if (idx_bytes != 1 && idx_bytes != 2 && idx_bytes != 4 && idx_bytes != 8) return fail(IVJ_EINVAL, "remap: idx_bytes must be 1, 2, 4 or 8");
    if (n == 0) return IVJ_OK;
    const int t = fd_threads(n, threads, 1 << 17);
    std::vector<std::vector<uint8_t>> sk(t, std::vector<uint8_t>((size_t)remap_len + 1, 0));
    std::vector<char> bad(t, 0);
    fd_parallel(n, t, [&](int k, int64_t lo, int64_t hi) {
        bool b = false;
        switch (idx_bytes) {
            case 8: fd_remap_range((const int64_t*)idx, lo, hi, remap, remap_len, out, sk[k].data(), b); break;
            case 4: fd_remap_range((const int32_t*)idx, lo, hi, remap, remap_len, out, sk[k].data(), b); break;
            case 2: fd_remap_range((const int16_t*)idx, lo, hi, remap, remap_len, out, sk[k].data(), b); break;
            default: fd_remap_range((const int8_t*)idx, lo, hi, remap, remap_len, out, sk[k].data(), b); break;
        }
        bad[k] = b;
    });
    for (int k = 0; k < t; ++k) {
        if (bad[k]) return fail(IVJ_EINVAL, "remap: an index lies outside the dictionary");
        for (int64_t v = 0; v < remap_len; ++v) seen[v] |= sk[k][v];
    }
    return IVJ_OK;
}

static int ivj_host_take_impl(const void* src, int32_t elem_bytes, int64_t n_src, const int32_t* idx, int64_t n, void* dst, int32_t threads) {
    if (n < 0 || n_src < 0 || (n > 0 && (!idx || !dst)) || (n_src > 0 && !src)) return fail(IVJ_EINVAL, "host take: bad argument");
    if (elem_bytes != 4 && elem_bytes != 8) return fail(IVJ_EINVAL, "host take: elem_bytes must be 4 or 8");
    if (n == 0) return IVJ_OK;
    const int t = fd_threads(n, threads, 1 << 15);
    fd_parallel(n, t, [&](int, int64_t lo, int64_t hi) {
        if (elem_bytes == 8) fd_take_range((const uint64_t*)src, n_src, idx, lo, hi, (uint64_t*)dst);
        else fd_take_range((const uint32_t*)src, n_src, idx, lo, hi, (uint32_t*)dst);
    });
    return IVJ_OK;
}

// dst[idx[i]] = src[i] (rows of row_bytes = 4, 8 or any other width): the mirror of ivj_host_take -- per-probe results of a shard back
// to their global rows.  The indices of one call are distinct by contract (a shard's rows), so the threads never meet; an index outside
// [0, n_dst) is refused before anything is written.  Optional remap (4-byte rows only): the stored value is remap[src[i]] for
// src[i] >= 0 and -1 otherwise (a shard's local build rows -> global build rows on the way).
static int ivj_host_scatter_impl(const void* src, int32_t row_bytes, int64_t n, const int32_t* idx, int64_t n_dst, void* dst, const int32_t* remap,
                                 int64_t remap_len, int32_t threads) {
    if (n < 0 || n_dst < 0 || row_bytes < 1 || (n > 0 && (!src || !idx || !dst))) return fail(IVJ_EINVAL, "host scatter: bad argument");
    if (remap && row_bytes % 4 != 0) return fail(IVJ_EINVAL, "host scatter: a remap needs rows of 4-byte values");
    if (n == 0) return IVJ_OK;
    const int t = fd_threads(n, threads, 1 << 15);
    std::vector<char> bad((size_t)t + 1, 0);
    fd_parallel(n, t, [&](int k, int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) if ((uint64_t)(int64_t)idx[i] >= (uint64_t)n_dst) { bad[(size_t)k] = 1; break; }
    });
    for (char b : bad) if (b) return fail(IVJ_EINVAL, "host scatter: a row index lies outside the destination");
    fd_parallel(n, t, [&](int, int64_t lo, int64_t hi) {
        if (remap) {
            const int w = row_bytes / 4;
            for (int64_t i = lo; i < hi; ++i)
                for (int j = 0; j < w; ++j) {
                    const int32_t v = ((const int32_t*)src)[i * w + j];
                    ((int32_t*)dst)[(int64_t)idx[i] * w + j] = (v >= 0 && (int64_t)v < remap_len) ? remap[v] : -1;
                }
        } else if (row_bytes == 8) {
            for (int64_t i = lo; i < hi; ++i) {
                if (i + 16 < hi) __builtin_prefetch((const char*)dst + (size_t)idx[i + 16] * 8, 1);
                ((uint64_t*)dst)[idx[i]] = ((const uint64_t*)src)[i];
            }
        } else if (row_bytes == 4) {
            for (int64_t i = lo; i < hi; ++i) {
                if (i + 16 < hi) __builtin_prefetch((const char*)dst + (size_t)idx[i + 16] * 4, 1);
                ((uint32_t*)dst)[idx[i]] = ((const uint32_t*)src)[i];
            }
        } else {
            for (int64_t i = lo; i < hi; ++i) std::memcpy((char*)dst + (size_t)idx[i] * (size_t)row_bytes, (const char*)src + (size_t)i * (size_t)row_bytes, (size_t)row_bytes);
        }
    });
    return IVJ_OK;
}

static int ivj_host_widen_i32_impl(const int32_t* src, int64_t n, int64_t* dst, int32_t threads) {
    if (n < 0 || (n > 0 && (!src || !dst))) return fail(IVJ_EINVAL, "widen: bad argument");
    if (n == 0) return IVJ_OK;
    const int t = fd_threads(n, threads, 1 << 17);
    fd_parallel(n, t, [&](int, int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) dst[i] = (int64_t)src[i]; });
    return IVJ_OK;
}


// Contig sharding of one side for `world` ranks: ONE counting pass and ONE placing pass over the rows, both threaded, input
// order kept inside every rank's share.  owner[c] = rank of contig c; a row whose contig lies outside [0, n_contigs) belongs to
// no rank.  counts_only: fill counts[world] and stop (the caller sizes the outputs from it).
static int ivj_host_shard_impl(const int32_t* contig, const int32_t* start, const int32_t* end, int64_t n, const int32_t* owner, int32_t n_contigs,
                               int32_t world, int64_t* counts, int32_t* const* out_contig, int32_t* const* out_start, int32_t* const* out_end,
                               int32_t* const* out_row, int32_t threads) {
    if (n < 0 || world < 1 || n_contigs < 0 || !counts || (n > 0 && !contig) || (n_contigs > 0 && !owner)) return fail(IVJ_EINVAL, "shard: bad argument");
    for (int32_t c = 0; c < n_contigs; ++c) if (owner[c] < 0 || owner[c] >= world) return fail(IVJ_EINVAL, "shard: an owner lies outside [0, world)");
    const bool place = out_contig || out_start || out_end || out_row;
    if (place && n > 0 && (!start || !end)) return fail(IVJ_EINVAL, "shard: start / end is NULL");
    const int t = fd_threads(n, threads, 1 << 16);
    std::vector<int64_t> cnt((size_t)t * (size_t)world, 0);
    fd_parallel(n, t, [&](int k, int64_t lo, int64_t hi) {
        int64_t* c = cnt.data() + (size_t)k * (size_t)world;
        for (int64_t i = lo; i < hi; ++i) {
            const uint32_t cc = (uint32_t)contig[i];
            if (cc < (uint32_t)n_contigs) ++c[owner[cc]];
        }
    });
    // exclusive prefix over the parts, per rank: part k of rank r writes from base[k][r]
    for (int32_t r = 0; r < world; ++r) {
        int64_t run = 0;
        for (int k = 0; k < t; ++k) { const int64_t v = cnt[(size_t)k * world + r]; cnt[(size_t)k * world + r] = run; run += v; }
        counts[r] = run;
    }
    if (!place) return IVJ_OK;
    for (int32_t r = 0; r < world; ++r)
        if (counts[r] > 0 && ((out_contig && !out_contig[r]) || (out_start && !out_start[r]) || (out_end && !out_end[r]) || (out_row && !out_row[r])))
            return fail(IVJ_EINVAL, "shard: an output column of a rank with rows is NULL");
    fd_parallel(n, t, [&](int k, int64_t lo, int64_t hi) {
        std::vector<int64_t> pos(cnt.begin() + (size_t)k * world, cnt.begin() + (size_t)(k + 1) * world);
        for (int64_t i = lo; i < hi; ++i) {
            const uint32_t cc = (uint32_t)contig[i];
            if (cc >= (uint32_t)n_contigs) continue;
            const int32_t r = owner[cc];
            const int64_t o = pos[(size_t)r]++;
            if (out_contig) out_contig[r][o] = (int32_t)cc;
            if (out_start) out_start[r][o] = start[i];
            if (out_end) out_end[r][o] = end[i];
            if (out_row) out_row[r][o] = (int32_t)i;
        }
    });
    return IVJ_OK;
}

// rows per contig of one side (the LPT weights of the contig -> rank assignment): hist[c] for c in [0, n_contigs)
static int ivj_host_contig_hist_impl(const int32_t* contig, int64_t n, int32_t n_contigs, int64_t* hist, int32_t threads) {
    if (n < 0 || n_contigs < 0 || (n > 0 && !contig) || (n_contigs > 0 && !hist)) return fail(IVJ_EINVAL, "contig hist: bad argument");
    for (int32_t c = 0; c < n_contigs; ++c) hist[c] = 0;
    if (n == 0 || n_contigs == 0) return IVJ_OK;
    const int t = fd_threads(n, threads, 1 << 17);
    std::vector<int64_t> part((size_t)t * (size_t)n_contigs, 0);
    fd_parallel(n, t, [&](int k, int64_t lo, int64_t hi) {
        int64_t* h = part.data() + (size_t)k * (size_t)n_contigs;
        for (int64_t i = lo; i < hi; ++i) { const uint32_t cc = (uint32_t)contig[i]; if (cc < (uint32_t)n_contigs) ++h[cc]; }
    });
    for (int k = 0; k < t; ++k) for (int32_t c = 0; c < n_contigs; ++c) hist[c] += part[(size_t)k * n_contigs + c];
    return IVJ_OK;
}


// no C++ exception may cross the C ABI (std::thread / std::vector can throw under resource exhaustion)
#define IVJ_HOST_GUARD(call)                                                                        \
    try { return call; }                                                                            \
    catch (const std::bad_alloc&) { return fail(IVJ_ENOMEM, "host helper: out of memory"); }        \
    catch (const std::exception& e) { return fail(IVJ_EINVAL, std::string("host helper: ") + e.what()); }

int ivj_host_narrow_i32(const void* src, int32_t src_bytes, int32_t is_unsigned, int64_t n, int32_t* dst, int64_t* out_min, int64_t* out_max,
                        int32_t threads) {
    IVJ_HOST_GUARD(ivj_host_narrow_i32_impl(src, src_bytes, is_unsigned, n, dst, out_min, out_max, threads))
}
int ivj_host_encode_utf8(const void* offsets, int32_t offset_bytes, const uint8_t* data, const uint8_t* validity, int64_t validity_bit0, int64_t n,
                         int32_t* ids, int64_t* dict_rows, int32_t dict_cap, int32_t* n_values, int32_t threads) {
    IVJ_HOST_GUARD(ivj_host_encode_utf8_impl(offsets, offset_bytes, data, validity, validity_bit0, n, ids, dict_rows, dict_cap, n_values, threads))
}
int ivj_host_encode_keys64(const uint64_t* keys, int64_t n, int32_t* ids, int64_t* dict_rows, int32_t dict_cap, int32_t* n_values, int32_t threads) {
    IVJ_HOST_GUARD(ivj_host_encode_keys64_impl(keys, n, ids, dict_rows, dict_cap, n_values, threads))
}
int ivj_host_remap_i32(const void* idx, int32_t idx_bytes, int64_t n, const int32_t* remap, int64_t remap_len, int32_t* out, uint8_t* seen,
                       int32_t threads) {
    IVJ_HOST_GUARD(ivj_host_remap_i32_impl(idx, idx_bytes, n, remap, remap_len, out, seen, threads))
}
int ivj_host_take(const void* src, int32_t elem_bytes, int64_t n_src, const int32_t* idx, int64_t n, void* dst, int32_t threads) {
    IVJ_HOST_GUARD(ivj_host_take_impl(src, elem_bytes, n_src, idx, n, dst, threads))
}
int ivj_host_scatter(const void* src, int32_t row_bytes, int64_t n, const int32_t* idx, int64_t n_dst, void* dst, const int32_t* remap, int64_t remap_len,
                     int32_t threads) {
    IVJ_HOST_GUARD(ivj_host_scatter_impl(src, row_bytes, n, idx, n_dst, dst, remap, remap_len, threads))
}
int ivj_host_widen_i32(const int32_t* src, int64_t n, int64_t* dst, int32_t threads) {
    IVJ_HOST_GUARD(ivj_host_widen_i32_impl(src, n, dst, threads))
}
int ivj_host_shard(const int32_t* contig, const int32_t* start, const int32_t* end, int64_t n, const int32_t* owner, int32_t n_contigs, int32_t world,
                   int64_t* counts, int32_t* const* out_contig, int32_t* const* out_start, int32_t* const* out_end, int32_t* const* out_row, int32_t threads) {
    IVJ_HOST_GUARD(ivj_host_shard_impl(contig, start, end, n, owner, n_contigs, world, counts, out_contig, out_start, out_end, out_row, threads))
}
int ivj_host_contig_hist(const int32_t* contig, int64_t n, int32_t n_contigs, int64_t* hist, int32_t threads) {
    IVJ_HOST_GUARD(ivj_host_contig_hist_impl(contig, n, n_contigs, hist, threads))
}
#undef IVJ_HOST_GUARD

}  // extern "C"
