// host_arrow_stream.hip.h -- the one-call Arrow entry: two ArrowArrayStreams in, a stream of joined record batches out
// Part of the single translation unit ivjoin.hip (included there, after host_frontdoor.hip.h); not a stand-alone header.
//
// What the reference's FFI binds is ONE call taking two Arrow C streams with a string chrom and start / end of any integer
// width, answering with a lazy frame of joined rows (/root/reference/src/lib.rs:79-145 range_operation_frame, :154-214
// range_operation_lazy; the renaming SELECT over the joined batches: src/operation.rs:272-301, 170-197).  The entry points
// here are that call for a C / Rust host:
//   1. both streams are drained into host batches (kept as they came: no concatenation of the payload columns);
//   2. the key columns are made: chrom (utf8 / large_utf8 / dictionary of those, per batch) -> ids of ONE dictionary shared
//      by both sides (the native hash pass of host_frontdoor.hip.h per batch, the few distinct values merged through a map),
//      start / end of any integer width -> int32 with the reference's range check (docs/features/operations.md:36-37);
//   3. the join runs through the host entry points of this library (ivj_overlap / ivj_count_overlaps / ivj_nearest:
//      H2D, HIP kernels, D2H) -- there is no CPU path;
//   4. the result is an ArrowArrayStream whose batches are assembled ON PULL: every column of df1 (+ suffix 1) and of df2
//      (+ suffix 2) gathered by the pair indices, `count` / `distance` appended for the per-probe operations.
// Column kinds the gather handles: fixed-width primitives of 1 .. 32 bytes (integers, floats, dates, times, timestamps,
// durations, decimals, fixed-size binary), bool, utf8 / large_utf8 / binary / large_binary, and dictionary-encoded columns of
// those (delivered DECODED to their value type: the dictionaries of different batches of one stream need not agree).
// Anything else (nested types, views, unions, run-end encoded) is refused by name when the call is made.
#pragma once

#include <memory>
#include <unordered_map>

// (ArrowSchema / ArrowArray: arrow_cdata.hip.h)
struct ArrowArrayStream {
    int (*get_schema)(struct ArrowArrayStream*, struct ArrowSchema* out);
    int (*get_next)(struct ArrowArrayStream*, struct ArrowArray* out);
    const char* (*get_last_error)(struct ArrowArrayStream*);
    void (*release)(struct ArrowArrayStream*);
    void* private_data;
};

namespace {

constexpr int64_t AS_NULLABLE = 2;                         // ARROW_FLAG_NULLABLE

enum AsKind { AS_FIXED, AS_BOOL, AS_STR32, AS_STR64, AS_UNSUPPORTED };
struct AsType {
    AsKind kind = AS_UNSUPPORTED;
    int width = 0;                                         // bytes of a fixed-width value
    bool dict = false;                                     // dictionary-encoded in the source (kind / width / format describe the VALUES)
    int idx_width = 0;                                     //   width of its indices
    bool idx_unsigned = false;
    std::string format;                                    // export format (the value type's)
};

int as_fixed_width(const char* f) {
    if (!f || !*f) return 0;
    switch (f[0]) {
        case 'c': case 'C': return f[1] ? 0 : 1;
        case 's': case 'S': case 'e': return f[1] ? 0 : 2;
        case 'i': case 'I': case 'f': return f[1] ? 0 : 4;
        case 'l': case 'L': case 'g': return f[1] ? 0 : 8;
        case 'w': return f[1] == ':' ? std::atoi(f + 2) : 0;                 // fixed-size binary
        case 'd': {                                                           // decimal "d:p,s[,bits]"
            if (f[1] != ':') return 0;
            int commas = 0; const char* last = nullptr;
            for (const char* p = f + 2; *p; ++p) if (*p == ',') { ++commas; last = p + 1; }
            if (commas < 2) return 16;
            const int bits = std::atoi(last);
            return bits % 8 == 0 ? bits / 8 : 0;
        }
        case 't':
            if (f[1] == 'd') return f[2] == 'D' ? 4 : (f[2] == 'm' ? 8 : 0);                  // date32 / date64
            if (f[1] == 't') return (f[2] == 's' || f[2] == 'm') ? 4 : ((f[2] == 'u' || f[2] == 'n') ? 8 : 0);   // time32 / time64
            if (f[1] == 's' || f[1] == 'D') return 8;                                         // timestamp / duration
            if (f[1] == 'i') return f[2] == 'M' ? 4 : (f[2] == 'D' ? 8 : (f[2] == 'n' ? 16 : 0));   // intervals
            return 0;
        default: return 0;
    }
}

AsType as_type_of(const ArrowSchema* s) {
    AsType t;
    const ArrowSchema* v = s;
    if (s->dictionary) {
        t.dict = true;
        const char* f = s->format ? s->format : "";
        t.idx_width = as_fixed_width(f);
        t.idx_unsigned = f[0] == 'C' || f[0] == 'S' || f[0] == 'I' || f[0] == 'L';
        if (!(t.idx_width == 1 || t.idx_width == 2 || t.idx_width == 4 || t.idx_width == 8) || f[1]) return t;
        v = s->dictionary;
        if (v->dictionary) return t;                        // a dictionary of dictionaries: refused
    }
    const char* f = v->format ? v->format : "";
    t.format = f;
    if (!std::strcmp(f, "b")) t.kind = AS_BOOL;
    else if (!std::strcmp(f, "u") || !std::strcmp(f, "z")) t.kind = AS_STR32;
    else if (!std::strcmp(f, "U") || !std::strcmp(f, "Z")) t.kind = AS_STR64;
    else {
        const int w = as_fixed_width(f);
        if (w >= 1 && w <= 32) { t.kind = AS_FIXED; t.width = w; }
    }
    return t;
}

inline bool as_bit(const uint8_t* bits, int64_t i) { return (bits[i >> 3] >> (i & 7)) & 1; }

// One drained stream: the schema and every batch as it came (owned; released with the table)
struct AsTable {
    ArrowSchema schema{};
    std::vector<ArrowArray> batches;
    std::vector<int64_t> start;                            // batches.size() + 1 row offsets
    std::vector<AsType> types;                             // per column
    int64_t n = 0;
    AsTable() { schema.release = nullptr; }
    AsTable(const AsTable&) = delete;
    ~AsTable() {
        for (ArrowArray& b : batches) if (b.release) b.release(&b);
        if (schema.release) schema.release(&schema);
    }
    int ncols() const { return (int)schema.n_children; }
    const char* name(int c) const { return schema.children[c]->name ? schema.children[c]->name : ""; }
    int find(const char* nm) const {
        for (int c = 0; c < ncols(); ++c) if (!std::strcmp(name(c), nm)) return c;
        return -1;
    }
    // batch holding global row r
    int batch_of(int64_t r) const {
        int lo = 0, hi = (int)batches.size() - 1;
        while (lo < hi) { const int m = (lo + hi + 1) >> 1; if (start[m] <= r) lo = m; else hi = m - 1; }
        return lo;
    }
};

// one batch against the stream's schema: the record-batch struct itself carries no nulls, every column has the buffers its type
// reads (a producer whose batch layout differs from its schema is refused by name, not dereferenced)
int as_check_batch(const ArrowSchema& schema, const std::vector<AsType>& types, const ArrowArray& a, const char* what) {
    if (a.n_children != schema.n_children) return fail(IVJ_EINVAL, std::string(what) + ": a batch disagrees with the schema on the number of columns");
    if (a.null_count > 0) return fail(IVJ_EINVAL, std::string(what) + ": a record batch with null ROWS (a validity bitmap on the struct itself) is not supported");
    for (int64_t c = 0; c < a.n_children; ++c) {
        const ArrowArray* ch = a.children ? a.children[c] : nullptr;
        const AsType& ty = types[(size_t)c];
        const char* nm = schema.children[c]->name ? schema.children[c]->name : "";
        if (!ch) return fail(IVJ_EINVAL, std::string(what) + ": column '" + nm + "' of a batch is missing");
        if (ty.kind == AS_UNSUPPORTED) continue;                                // refused later, by name, if the result needs it
        const int need = ty.dict ? 2 : (ty.kind == AS_STR32 || ty.kind == AS_STR64 ? 3 : 2);
        if (ch->n_buffers < need || !ch->buffers) return fail(IVJ_EINVAL, std::string(what) + ": column '" + nm + "' of a batch has " + std::to_string(ch->n_buffers) + " buffers, its type needs " + std::to_string(need));
        if (ch->length + ch->offset > 0 && !ch->buffers[1]) return fail(IVJ_EINVAL, std::string(what) + ": column '" + nm + "' of a batch has no data buffer");
        if (ty.dict && !ch->dictionary) return fail(IVJ_EINVAL, std::string(what) + ": a batch of dictionary column '" + nm + "' carries no dictionary");
        if (ch->length < a.length + a.offset - ch->offset && ch->length < a.length) return fail(IVJ_EINVAL, std::string(what) + ": column '" + nm + "' of a batch is shorter than the batch");
    }
    return IVJ_OK;
}

// The stream is CONSUMED whatever happens: on success and on every failure the producer's release callback has run when this
// returns (a caller that gets an error back owns nothing any more -- the contract of the *_arrow_stream entry points).
int as_drain(ArrowArrayStream* in, const char* what, AsTable& t) {
    if (!in) return fail(IVJ_EINVAL, std::string(what) + ": not an ArrowArrayStream");
    struct Consume { ArrowArrayStream* s; ~Consume() { if (s->release) s->release(s); } } consume{in};      // (installed before ANY early return)
    if (!in->get_schema || !in->get_next) return fail(IVJ_EINVAL, std::string(what) + ": not an ArrowArrayStream");
    auto err = [&](const char* step) {
        const char* m = in->get_last_error ? in->get_last_error(in) : nullptr;
        return fail(IVJ_EINVAL, std::string(what) + ": " + step + " failed" + (m ? std::string(": ") + m : std::string()));
    };
    if (in->get_schema(in, &t.schema) != 0) return err("get_schema");
    if (!t.schema.format || std::strcmp(t.schema.format, "+s") != 0) return fail(IVJ_EINVAL, std::string(what) + ": the stream's schema is not a struct (record batches expected)");
    t.types.resize((size_t)t.ncols());
    for (int c = 0; c < t.ncols(); ++c) t.types[(size_t)c] = as_type_of(t.schema.children[c]);
    t.start.push_back(0);
    for (;;) {
        ArrowArray a{};
        a.release = nullptr;
        if (in->get_next(in, &a) != 0) return err("get_next");
        if (!a.release) break;                              // end of stream
        const int rc = as_check_batch(t.schema, t.types, a, what);
        if (rc != IVJ_OK) { a.release(&a); return rc; }
        if (a.length == 0) { a.release(&a); continue; }
        t.batches.push_back(a);
        t.n += a.length;
        t.start.push_back(t.n);
    }
    if (t.n > (int64_t)INT32_MAX) return fail(IVJ_EINVAL, std::string(what) + ": more than 2^31 - 1 rows on one side");
    return IVJ_OK;
}

// ---- key columns ---------------------------------------------------------------------------------------------------------------
struct AsDict {                                            // the chrom dictionary shared by both sides
    std::unordered_map<std::string, int32_t> map;
    std::vector<std::string> names;
    int32_t id_of(const char* p, int64_t len) {
        std::string s(p, (size_t)len);
        auto it = map.find(s);
        if (it != map.end()) return it->second;
        const int32_t id = (int32_t)names.size();
        map.emplace(s, id);
        names.push_back(std::move(s));
        return id;
    }
};

// string values of one array (offset applied by the caller through `first`): value i = [off[first + i], off[first + i + 1])
struct AsStrView {
    const void* off; int off_bytes; const uint8_t* data; const uint8_t* valid; int64_t first;
    int64_t begin(int64_t i) const { return off_bytes == 4 ? (int64_t)((const int32_t*)off)[first + i] : ((const int64_t*)off)[first + i]; }
    int64_t end(int64_t i) const { return off_bytes == 4 ? (int64_t)((const int32_t*)off)[first + i + 1] : ((const int64_t*)off)[first + i + 1]; }
    bool is_valid(int64_t i) const { return !valid || as_bit(valid, first + i); }
};
AsStrView as_str_view(const ArrowArray* a, int64_t extra_offset, int off_bytes) {
    static const uint8_t none = 0;
    AsStrView v;
    v.off = a->buffers[1]; v.off_bytes = off_bytes;
    v.data = a->n_buffers > 2 && a->buffers[2] ? (const uint8_t*)a->buffers[2] : &none;
    v.valid = (a->null_count != 0 && a->buffers[0]) ? (const uint8_t*)a->buffers[0] : nullptr;
    v.first = a->offset + extra_offset;
    return v;
}

int as_encode_chrom(const AsTable& t, int col, const char* what, AsDict& dict, int32_t* ids, int threads) {
    const AsType& ty = t.types[(size_t)col];
    if (!(ty.kind == AS_STR32 || ty.kind == AS_STR64) || (ty.format != "u" && ty.format != "U"))
        return fail(IVJ_EINVAL, std::string(what) + ": column '" + t.name(col) + "' must be utf8, large_utf8 or a dictionary of those (format '" +
                                    (t.schema.children[col]->format ? t.schema.children[col]->format : "") + "')");
    const int ob = ty.kind == AS_STR32 ? 4 : 8;
    std::vector<int64_t> rows((size_t)FD_MAX_VALUES);
    for (size_t b = 0; b < t.batches.size(); ++b) {
        const ArrowArray& top = t.batches[b];
        const ArrowArray* a = top.children[col];
        const int64_t n = top.length;
        int32_t* out = ids + t.start[b];
        if (ty.dict) {
            const ArrowArray* d = a->dictionary;
            if (!d) return fail(IVJ_EINVAL, std::string(what) + ": a batch of dictionary column '" + t.name(col) + "' carries no dictionary");
            const AsStrView dv = as_str_view(d, 0, ob);
            std::vector<int32_t> remap((size_t)d->length);
            for (int64_t v = 0; v < d->length; ++v) remap[(size_t)v] = dv.is_valid(v) ? dict.id_of((const char*)dv.data + dv.begin(v), dv.end(v) - dv.begin(v)) : -1;
            const uint8_t* valid = (a->null_count != 0 && a->buffers[0]) ? (const uint8_t*)a->buffers[0] : nullptr;
            const int64_t first = a->offset + top.offset;
            bool bad = false;
            const int tn = fd_threads(n, threads, 1 << 15);
            std::vector<char> badk((size_t)tn, 0);
            fd_parallel(n, tn, [&](int k, int64_t lo, int64_t hi) {
                for (int64_t i = lo; i < hi; ++i) {
                    long long v;
                    switch (ty.idx_width) {
                        case 1: v = ty.idx_unsigned ? (long long)((const uint8_t*)a->buffers[1])[first + i] : (long long)((const int8_t*)a->buffers[1])[first + i]; break;
                        case 2: v = ty.idx_unsigned ? (long long)((const uint16_t*)a->buffers[1])[first + i] : (long long)((const int16_t*)a->buffers[1])[first + i]; break;
                        case 4: v = ty.idx_unsigned ? (long long)((const uint32_t*)a->buffers[1])[first + i] : (long long)((const int32_t*)a->buffers[1])[first + i]; break;
                        default: v = (long long)((const int64_t*)a->buffers[1])[first + i]; break;
                    }
                    if (valid && !as_bit(valid, first + i)) { out[i] = -1; continue; }
                    if (v < 0 || v >= (long long)d->length) { badk[(size_t)k] = 1; out[i] = -1; continue; }
                    out[i] = remap[(size_t)v];
                }
            });
            for (char x : badk) bad = bad || x;
            if (bad) return fail(IVJ_EINVAL, std::string(what) + ": a dictionary index of column '" + t.name(col) + "' lies outside its dictionary");
            continue;
        }
        const AsStrView sv = as_str_view(a, top.offset, ob);
        int32_t nv = 0;
        const void* off0 = (const char*)sv.off + (size_t)sv.first * (size_t)ob;
        const int rc = ivj_host_encode_utf8(off0, ob, sv.data, sv.valid, sv.first, n, out, rows.data(), FD_MAX_VALUES, &nv, threads);
        if (rc == IVJ_OK) {
            std::vector<int32_t> remap((size_t)nv);
            bool identity = true;
            for (int32_t v = 0; v < nv; ++v) {
                const int64_t r = rows[(size_t)v];
                remap[(size_t)v] = dict.id_of((const char*)sv.data + sv.begin(r), sv.end(r) - sv.begin(r));
                identity = identity && remap[(size_t)v] == v;
            }
            if (!identity) {
                const int tn = fd_threads(n, threads, 1 << 15);
                fd_parallel(n, tn, [&](int, int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) if (out[i] >= 0) out[i] = remap[(size_t)out[i]]; });
            }
        } else if (rc == IVJ_ECAPACITY) {                   // thousands of distinct names in one batch: the plain map, row by row
            for (int64_t i = 0; i < n; ++i) out[i] = sv.is_valid(i) ? dict.id_of((const char*)sv.data + sv.begin(i), sv.end(i) - sv.begin(i)) : -1;
        } else return rc;
    }
    return IVJ_OK;
}

int as_narrow_coord(const AsTable& t, int col, const char* what, int32_t* out, int threads) {
    const ArrowSchema* cs = t.schema.children[col];
    const char* f = cs->format ? cs->format : "";
    const bool ok = !cs->dictionary && f[0] && !f[1] && std::strchr("cCsSiIlL", f[0]);
    if (!ok) return fail(IVJ_EINVAL, std::string(what) + ": column '" + t.name(col) + "' must be an integer type, got format '" + f + "'");
    const int w = as_fixed_width(f);
    const int uns = f[0] == 'C' || f[0] == 'S' || f[0] == 'I' || f[0] == 'L';
    for (size_t b = 0; b < t.batches.size(); ++b) {
        const ArrowArray& top = t.batches[b];
        const ArrowArray* a = top.children[col];
        if (a->null_count != 0) {
            // (null_count -1 = unknown: count the bitmap)
            bool any = a->null_count > 0;
            if (a->null_count < 0 && a->buffers[0])
                for (int64_t i = 0; i < top.length && !any; ++i) any = !as_bit((const uint8_t*)a->buffers[0], a->offset + top.offset + i);
            if (any) return fail(IVJ_EINVAL, std::string(what) + ": column '" + t.name(col) + "' contains nulls; interval coordinates must be non-null");
        }
        int64_t mn = 0, mx = 0;
        const char* src = (const char*)a->buffers[1] + (size_t)(a->offset + top.offset) * (size_t)w;
        IVJ_TRY(ivj_host_narrow_i32(src, w, uns, top.length, out + t.start[b], &mn, &mx, threads));
        if (top.length > 0 && (mn < (int64_t)INT32_MIN || mx > (int64_t)INT32_MAX))
            return fail(IVJ_EINVAL, std::string(what) + ": column '" + t.name(col) + "' does not fit int32 coordinates (reference limit): Integer value " +
                                        std::to_string(mx > (int64_t)INT32_MAX ? mx : mn) + " not in range: -2147483648 to 2147483647");
    }
    return IVJ_OK;
}

struct AsKeys {
    std::vector<int32_t> c1, s1, e1, c2, s2, e2;
    AsDict dict;
};

int as_make_keys(const AsTable& t1, const AsTable& t2, const char* const* cols1, const char* const* cols2, AsKeys& K, int threads) {
    static const char* const dflt[3] = {"chrom", "start", "end"};
    const char* const* n1 = cols1 ? cols1 : dflt;
    const char* const* n2 = cols2 ? cols2 : dflt;
    int i1[3], i2[3];
    for (int k = 0; k < 3; ++k) {
        if (!n1[k] || !n2[k]) return fail(IVJ_EINVAL, "a key column name is NULL");
        i1[k] = t1.find(n1[k]); i2[k] = t2.find(n2[k]);
        if (i1[k] < 0) return fail(IVJ_EINVAL, std::string("df1: column '") + n1[k] + "' not found");
        if (i2[k] < 0) return fail(IVJ_EINVAL, std::string("df2: column '") + n2[k] + "' not found");
    }
    K.c1.resize((size_t)t1.n); K.s1.resize((size_t)t1.n); K.e1.resize((size_t)t1.n);
    K.c2.resize((size_t)t2.n); K.s2.resize((size_t)t2.n); K.e2.resize((size_t)t2.n);
    IVJ_TRY(as_encode_chrom(t1, i1[0], "df1", K.dict, K.c1.data(), threads));
    IVJ_TRY(as_encode_chrom(t2, i2[0], "df2", K.dict, K.c2.data(), threads));
    IVJ_TRY(as_narrow_coord(t1, i1[1], "df1", K.s1.data(), threads));
    IVJ_TRY(as_narrow_coord(t1, i1[2], "df1", K.e1.data(), threads));
    IVJ_TRY(as_narrow_coord(t2, i2[1], "df2", K.s2.data(), threads));
    IVJ_TRY(as_narrow_coord(t2, i2[2], "df2", K.e2.data(), threads));
    return IVJ_OK;
}

// ---- export: batches assembled on pull -------------------------------------------------------------------------------------------
struct AsOutCol {
    int side;                                              // 0 / 1: a column of df1 / df2 gathered by the side's row index; 2: an appended int64 column
    int col;
    std::string name;
    AsType type;
    bool from_ids = false;                                 // the chrom KEY column: rebuilt from the side's dictionary ids (AsResult::chrom_ids) and the
                                                           // few names of the shared dictionary instead of a string gather out of the 10^7-row input column
};

struct AsResult {
    std::shared_ptr<AsTable> t[2];
    std::vector<AsOutCol> cols;
    // row indices of the result (one per output row and side; -1 = null row), and the appended int64 column
    std::vector<int32_t> own_idx[2];
    const int32_t* idx[2] = {nullptr, nullptr};            // nullptr: identity (row r of the side)
    ivj_pairs pairs{0, nullptr, nullptr};                  // overlap: the library-owned pair buffers idx[] point into
    std::vector<int32_t> chrom_ids[2];                     // per row of the side: id in `names` (-1: null chrom)
    std::vector<std::string> names;                        // the shared chrom dictionary
    // ... or borrowed from a longer-lived owner (the lazy entry: one AsResult per probe batch over the session's dictionary)
    const int32_t* cid[2] = {nullptr, nullptr};
    int64_t cid_n[2] = {0, 0};
    const std::vector<std::string>* names_ref = nullptr;
    std::vector<int64_t> extra;
    std::vector<uint8_t> extra_null;                       // 1 = null (distance of a probe row without a neighbour)
    std::string extra_name;
    int64_t n_rows = 0, cursor = 0, batch_rows = 1 << 20;
    int threads = 0;
    std::string last_error;
    std::mutex mu;
    ~AsResult() { ivj_pairs_free(&pairs); }
};

// buffer of one result column: from 2 MiB on, huge-page aligned and advised (a fresh 32-MB column is 8192 first-touch faults of 4 KiB
// otherwise -- a quarter of the assembly time of a 4 M-row batch); free with std::free
inline void* as_alloc(size_t bytes) {
    const size_t huge = (size_t)2 << 20;
    if (bytes < huge) return std::malloc(bytes ? bytes : 1);
    void* p = nullptr;
    const size_t sz = (bytes + huge - 1) / huge * huge;
    if (posix_memalign(&p, huge, sz) != 0) return nullptr;
    (void)madvise(p, sz, MADV_HUGEPAGE);
    return p;
}

struct AsBufOwner {                                        // one exported column: its buffers are malloc'ed and freed with it
    void* bufs[3] = {nullptr, nullptr, nullptr};
    const void* ptrs[3] = {nullptr, nullptr, nullptr};
    AsBufOwner() = default;
    AsBufOwner(const AsBufOwner&) = delete;
    ~AsBufOwner() { for (void* p : bufs) std::free(p); }   // (also on the error paths of the column builders: the guard deletes the owner)
};
void as_release_col(ArrowArray* a) {
    delete static_cast<AsBufOwner*>(a->private_data);
    a->release = nullptr;
}
struct AsBatchOwner { std::vector<ArrowArray> child; std::vector<ArrowArray*> ptrs; const void* top_buf[1] = {nullptr}; };
void as_release_batch(ArrowArray* a) {
    auto* o = static_cast<AsBatchOwner*>(a->private_data);
    for (ArrowArray& c : o->child) if (c.release) c.release(&c);
    delete o;
    a->release = nullptr;
}
struct AsSchemaOwner { std::vector<ArrowSchema> child; std::vector<ArrowSchema*> ptrs; std::vector<std::string> names, formats; };
void as_release_schema_child(ArrowSchema* s) { s->release = nullptr; }
void as_release_schema(ArrowSchema* s) {
    auto* o = static_cast<AsSchemaOwner*>(s->private_data);
    delete o;
    s->release = nullptr;
}

// (batch, local row) of every requested row of one side for output rows [lo, hi); local = -1 for a null row
struct AsLoc { std::vector<int32_t> batch, row; };
void as_locate(const AsTable& t, const int32_t* idx, int64_t lo, int64_t hi, AsLoc& L, int threads) {
    const int64_t n = hi - lo;
    L.batch.resize((size_t)n); L.row.resize((size_t)n);
    const int tn = fd_threads(n, threads, 1 << 15);
    fd_parallel(n, tn, [&](int, int64_t a, int64_t b) {
        int last = 0;
        for (int64_t i = a; i < b; ++i) {
            const int64_t r = idx ? (int64_t)idx[lo + i] : lo + i;
            if (r < 0 || r >= t.n) { L.batch[(size_t)i] = 0; L.row[(size_t)i] = -1; continue; }
            if (!(t.start[(size_t)last] <= r && r < t.start[(size_t)last + 1])) last = t.batch_of(r);
            L.batch[(size_t)i] = last;
            L.row[(size_t)i] = (int32_t)(r - t.start[(size_t)last]);
        }
    });
}

// source view of column `col` in batch b: value array + the index to add to a local row
struct AsSrc { const ArrowArray* a; int64_t first; const uint8_t* valid; const ArrowArray* d; };
inline AsSrc as_src(const AsTable& t, int b, int col) {
    const ArrowArray& top = t.batches[(size_t)b];
    const ArrowArray* a = top.children[col];
    AsSrc s;
    s.a = a; s.first = a->offset + top.offset;
    s.valid = (a->null_count != 0 && a->buffers[0]) ? (const uint8_t*)a->buffers[0] : nullptr;
    s.d = a->dictionary;
    return s;
}
inline long long as_dict_index(const AsType& ty, const AsSrc& s, int64_t i) {
    const void* p = s.a->buffers[1];
    switch (ty.idx_width) {
        case 1: return ty.idx_unsigned ? (long long)((const uint8_t*)p)[s.first + i] : (long long)((const int8_t*)p)[s.first + i];
        case 2: return ty.idx_unsigned ? (long long)((const uint16_t*)p)[s.first + i] : (long long)((const int16_t*)p)[s.first + i];
        case 4: return ty.idx_unsigned ? (long long)((const uint32_t*)p)[s.first + i] : (long long)((const int32_t*)p)[s.first + i];
        default: return (long long)((const int64_t*)p)[s.first + i];
    }
}
// resolves (batch, local row) to the array that holds the VALUE and the value's index there; false = null
inline bool as_resolve(const AsTable& t, const AsType& ty, int col, int b, int32_t r, const ArrowArray*& va, int64_t& vi) {
    if (r < 0) return false;
    const AsSrc s = as_src(t, b, col);
    if (s.valid && !as_bit(s.valid, s.first + r)) return false;
    if (!ty.dict) { va = s.a; vi = s.first + r; return true; }
    const long long v = as_dict_index(ty, s, r);
    if (!s.d || v < 0 || v >= (long long)s.d->length) return false;
    vi = s.d->offset + v;
    va = s.d;
    if (s.d->null_count != 0 && s.d->buffers[0] && !as_bit((const uint8_t*)s.d->buffers[0], vi)) return false;
    return true;
}

// the chrom key column of output rows [lo, lo + n): names[ids[row]] -- offsets from the name lengths, bytes from the dictionary
int as_chrom_col(const AsResult& R, const AsOutCol& oc, int64_t lo, int64_t n, int threads, ArrowArray* out) {
    const int32_t* ids = R.cid[oc.side] ? R.cid[oc.side] : R.chrom_ids[oc.side].data();
    const std::vector<std::string>& names = R.names_ref ? *R.names_ref : R.names;
    const int32_t* idx = R.idx[oc.side];
    const int ob = oc.type.kind == AS_STR32 ? 4 : 8;
    auto* own = new AsBufOwner();
    std::unique_ptr<AsBufOwner> guard(own);
    uint8_t* valid = (uint8_t*)std::calloc((size_t)((n + 7) / 8) + 1, 1);
    int32_t* id = (int32_t*)std::malloc(n ? (size_t)n * 4 : 4);
    if (!valid || !id) { std::free(valid); std::free(id); return fail(IVJ_ENOMEM, "result batch: out of memory"); }
    own->bufs[0] = valid;
    std::unique_ptr<int32_t, void (*)(void*)> idg(id, std::free);
    std::vector<int64_t> nlen(names.size());
    for (size_t v = 0; v < names.size(); ++v) nlen[v] = (int64_t)names[v].size();
    const int tn = fd_threads(n, threads, 1 << 15);
    std::vector<int64_t> part((size_t)tn + 1, 0), nulls((size_t)tn, 0);
    const int64_t n_src = R.cid[oc.side] ? R.cid_n[oc.side] : (int64_t)R.chrom_ids[oc.side].size();
    fd_parallel(n, tn, [&](int k, int64_t a, int64_t b) {
        int64_t bytes = 0, nn = 0;
        for (int64_t i = a; i < b; ++i) {
            const int64_t r = idx ? (int64_t)idx[lo + i] : lo + i;
            int32_t v = (r >= 0 && r < n_src) ? ids[(size_t)r] : -1;
            if (v >= (int32_t)names.size()) v = -1;
            id[i] = v;
            if (v >= 0) { bytes += nlen[(size_t)v]; valid[i >> 3] |= (uint8_t)(1u << (i & 7)); } else ++nn;
        }
        part[(size_t)k + 1] = bytes; nulls[(size_t)k] = nn;
    });
    for (int k = 0; k < tn; ++k) part[(size_t)k + 1] += part[(size_t)k];
    const int64_t total = part[(size_t)tn];
    if (ob == 4 && total > (int64_t)INT32_MAX) return fail(IVJ_EINVAL, "result batch: the values of utf8 column '" + oc.name + "' pass 2 GiB in one batch; lower batch_rows");
    void* offs = as_alloc((size_t)(n + 1) * (size_t)ob);
    char* bytes = (char*)as_alloc((size_t)total);
    if (!offs || !bytes) { std::free(offs); std::free(bytes); return fail(IVJ_ENOMEM, "result batch: out of memory"); }
    own->bufs[1] = offs; own->bufs[2] = bytes;
    fd_parallel(n, tn, [&](int k, int64_t a, int64_t b) {
        int64_t o = part[(size_t)k];
        for (int64_t i = a; i < b; ++i) {
            if (ob == 4) ((int32_t*)offs)[i] = (int32_t)o; else ((int64_t*)offs)[i] = o;
            const int32_t v = id[i];
            if (v >= 0) { std::memcpy(bytes + o, names[(size_t)v].data(), (size_t)nlen[(size_t)v]); o += nlen[(size_t)v]; }
        }
    });
    if (ob == 4) ((int32_t*)offs)[n] = (int32_t)total; else ((int64_t*)offs)[n] = total;
    int64_t nn = 0;
    for (int64_t x : nulls) nn += x;
    for (int k = 0; k < 3; ++k) own->ptrs[k] = own->bufs[k];
    if (nn == 0) own->ptrs[0] = nullptr;
    *out = ArrowArray{n, nn, 0, 3, 0, own->ptrs, nullptr, nullptr, as_release_col, own};
    guard.release();
    return IVJ_OK;
}

int as_gather_col(const AsTable& t, const AsOutCol& oc, const AsLoc& L, int64_t n, int threads, ArrowArray* out) {
    const AsType& ty = oc.type;
    auto* own = new AsBufOwner();
    std::unique_ptr<AsBufOwner> guard(own);
    const size_t vbytes = (size_t)((n + 7) / 8);
    uint8_t* valid = (uint8_t*)std::calloc(vbytes ? vbytes : 1, 1);
    if (!valid) return fail(IVJ_ENOMEM, "result batch: out of memory");
    own->bufs[0] = valid;
    const int tn = fd_threads(n, threads, 1 << 14);
    std::vector<int64_t> nulls((size_t)tn, 0);
    int n_buffers = 2;
    // (every worker owns whole validity bytes: fd_parallel cuts at multiples of 64 rows)
    if (ty.kind == AS_FIXED && !ty.dict && (ty.width == 4 || ty.width == 8)) {
        // the common column (int32 / int64 / float / timestamp ...): the per-batch source views are made ONCE, the row loop is a typed
        // indexed copy, and the validity bitmap starts all-valid and is only touched for a null (round 5: the generic path below
        // resolved batch, offsets and dictionary per ROW -- 4 M rows x 8 columns cost 0.05 s where this costs 0.01 s)
        const size_t w = (size_t)ty.width;
        char* vals = (char*)as_alloc((size_t)n * w);
        if (!vals) return fail(IVJ_ENOMEM, "result batch: out of memory");
        own->bufs[1] = vals;
        std::memset(valid, 0xff, vbytes);
        if (n & 7) valid[vbytes - 1] = (uint8_t)((1u << (n & 7)) - 1u);         // (bits past the length stay clear)
        std::vector<AsSrc> srcs(t.batches.size());
        for (size_t b = 0; b < t.batches.size(); ++b) srcs[b] = as_src(t, (int)b, oc.col);
        const bool one = t.batches.size() == 1;
        fd_parallel(n, tn, [&](int k, int64_t lo, int64_t hi) {
            int64_t nn = 0;
            for (int64_t i = lo; i < hi; ++i) {
                const int32_t r = L.row[(size_t)i];
                const AsSrc& sv = srcs[one ? 0 : (size_t)L.batch[(size_t)i]];
                if (r < 0 || (sv.valid && !as_bit(sv.valid, sv.first + r))) {
                    if (w == 4) ((uint32_t*)vals)[i] = 0u; else ((uint64_t*)vals)[i] = 0ull;
                    valid[i >> 3] &= (uint8_t)~(1u << (i & 7));
                    ++nn;
                } else if (w == 4) ((uint32_t*)vals)[i] = ((const uint32_t*)sv.a->buffers[1])[sv.first + r];
                else ((uint64_t*)vals)[i] = ((const uint64_t*)sv.a->buffers[1])[sv.first + r];
            }
            nulls[(size_t)k] = nn;
        });
    } else if (ty.kind == AS_FIXED) {
        const size_t w = (size_t)ty.width;
        char* vals = (char*)as_alloc((size_t)n * w);
        if (!vals) return fail(IVJ_ENOMEM, "result batch: out of memory");
        own->bufs[1] = vals;
        fd_parallel(n, tn, [&](int k, int64_t lo, int64_t hi) {
            int64_t nn = 0;
            for (int64_t i = lo; i < hi; ++i) {
                const ArrowArray* va; int64_t vi;
                if (as_resolve(t, ty, oc.col, L.batch[(size_t)i], L.row[(size_t)i], va, vi)) {
                    std::memcpy(vals + (size_t)i * w, (const char*)va->buffers[1] + (size_t)vi * w, w);
                    valid[i >> 3] |= (uint8_t)(1u << (i & 7));
                } else { std::memset(vals + (size_t)i * w, 0, w); ++nn; }
            }
            nulls[(size_t)k] = nn;
        });
    } else if (ty.kind == AS_BOOL) {
        uint8_t* vals = (uint8_t*)std::calloc(vbytes ? vbytes : 1, 1);
        if (!vals) return fail(IVJ_ENOMEM, "result batch: out of memory");
        own->bufs[1] = vals;
        fd_parallel(n, tn, [&](int k, int64_t lo, int64_t hi) {
            int64_t nn = 0;
            for (int64_t i = lo; i < hi; ++i) {
                const ArrowArray* va; int64_t vi;
                if (as_resolve(t, ty, oc.col, L.batch[(size_t)i], L.row[(size_t)i], va, vi)) {
                    if (as_bit((const uint8_t*)va->buffers[1], vi)) vals[i >> 3] |= (uint8_t)(1u << (i & 7));
                    valid[i >> 3] |= (uint8_t)(1u << (i & 7));
                } else ++nn;
            }
            nulls[(size_t)k] = nn;
        });
    } else {                                                // strings / binary: lengths -> offsets -> bytes
        const int ob = ty.kind == AS_STR32 ? 4 : 8;
        n_buffers = 3;
        std::vector<int64_t> len((size_t)n + 1, 0);
        std::vector<const char*> src((size_t)n, nullptr);
        fd_parallel(n, tn, [&](int k, int64_t lo, int64_t hi) {
            int64_t nn = 0;
            for (int64_t i = lo; i < hi; ++i) {
                const ArrowArray* va; int64_t vi;
                if (as_resolve(t, ty, oc.col, L.batch[(size_t)i], L.row[(size_t)i], va, vi)) {
                    int64_t a, b;
                    if (ob == 4) { a = ((const int32_t*)va->buffers[1])[vi]; b = ((const int32_t*)va->buffers[1])[vi + 1]; }
                    else { a = ((const int64_t*)va->buffers[1])[vi]; b = ((const int64_t*)va->buffers[1])[vi + 1]; }
                    len[(size_t)i] = b - a;
                    src[(size_t)i] = (const char*)va->buffers[2] + a;
                    valid[i >> 3] |= (uint8_t)(1u << (i & 7));
                } else ++nn;
            }
            nulls[(size_t)k] = nn;
        });
        int64_t total = 0;
        for (int64_t i = 0; i < n; ++i) { const int64_t l = len[(size_t)i]; len[(size_t)i] = total; total += l; }
        len[(size_t)n] = total;
        if (ob == 4 && total > (int64_t)INT32_MAX)
            return fail(IVJ_EINVAL, "result batch: the values of utf8 column '" + oc.name + "' pass 2 GiB in one batch; lower batch_rows or hand the column over as large_utf8");
        void* offs = as_alloc((size_t)(n + 1) * (size_t)ob);
        char* bytes = (char*)as_alloc((size_t)total);
        if (!offs || !bytes) { std::free(offs); std::free(bytes); return fail(IVJ_ENOMEM, "result batch: out of memory"); }
        own->bufs[1] = offs; own->bufs[2] = bytes;
        fd_parallel(n + 1, fd_threads(n + 1, threads, 1 << 15), [&](int, int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) { if (ob == 4) ((int32_t*)offs)[i] = (int32_t)len[(size_t)i]; else ((int64_t*)offs)[i] = len[(size_t)i]; }
        });
        fd_parallel(n, tn, [&](int, int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) if (src[(size_t)i]) std::memcpy(bytes + len[(size_t)i], src[(size_t)i], (size_t)(len[(size_t)i + 1] - len[(size_t)i]));
        });
    }
    int64_t nn = 0;
    for (int64_t x : nulls) nn += x;
    for (int k = 0; k < 3; ++k) own->ptrs[k] = own->bufs[k];
    if (nn == 0) own->ptrs[0] = nullptr;                    // no nulls: the bitmap may be omitted (it is still freed)
    *out = ArrowArray{n, nn, 0, n_buffers, 0, own->ptrs, nullptr, nullptr, as_release_col, own};
    guard.release();
    return IVJ_OK;
}

int as_extra_col(const AsResult& R, int64_t lo, int64_t n, ArrowArray* out) {
    auto* own = new AsBufOwner();
    std::unique_ptr<AsBufOwner> guard(own);
    int64_t* vals = (int64_t*)as_alloc((size_t)n * 8);
    uint8_t* valid = (uint8_t*)std::calloc((size_t)((n + 7) / 8) + 1, 1);
    if (!vals || !valid) { std::free(vals); std::free(valid); return fail(IVJ_ENOMEM, "result batch: out of memory"); }
    own->bufs[0] = valid; own->bufs[1] = vals;
    int64_t nn = 0;
    for (int64_t i = 0; i < n; ++i) {
        const bool null = !R.extra_null.empty() && R.extra_null[(size_t)(lo + i)];
        vals[i] = null ? 0 : R.extra[(size_t)(lo + i)];
        if (null) ++nn; else valid[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
    own->ptrs[0] = nn ? valid : nullptr; own->ptrs[1] = vals;
    *out = ArrowArray{n, nn, 0, 2, 0, own->ptrs, nullptr, nullptr, as_release_col, own};
    guard.release();
    return IVJ_OK;
}

int as_result_schema(const AsResult& R, ArrowSchema* out) {
    auto* o = new AsSchemaOwner();
    const size_t nc = R.cols.size();
    o->child.resize(nc); o->ptrs.resize(nc); o->names.resize(nc); o->formats.resize(nc);
    for (size_t c = 0; c < nc; ++c) {
        o->names[c] = R.cols[c].name;
        o->formats[c] = R.cols[c].side == 2 ? std::string("l") : R.cols[c].type.format;
    }
    for (size_t c = 0; c < nc; ++c) {
        o->child[c] = ArrowSchema{o->formats[c].c_str(), o->names[c].c_str(), nullptr, AS_NULLABLE, 0, nullptr, nullptr, as_release_schema_child, nullptr};
        o->ptrs[c] = &o->child[c];
    }
    *out = ArrowSchema{"+s", "", nullptr, 0, (int64_t)nc, o->ptrs.data(), nullptr, as_release_schema, o};
    return IVJ_OK;
}

int as_stream_get_schema(ArrowArrayStream* s, ArrowSchema* out) {
    auto* R = static_cast<AsResult*>(s->private_data);
    try { return as_result_schema(*R, out) == IVJ_OK ? 0 : EINVAL; }
    catch (const std::exception& e) { R->last_error = e.what(); return ENOMEM; }
}

// the next batch_rows rows of R as one record batch (out->release == nullptr: R is exhausted); 0 or an errno, R->last_error set
int as_next_batch(AsResult* R, ArrowArray* out) {
    try {
        if (R->cursor >= R->n_rows) { std::memset(out, 0, sizeof(*out)); out->release = nullptr; return 0; }     // end of stream
        const int64_t lo = R->cursor, hi = lo + R->batch_rows < R->n_rows ? lo + R->batch_rows : R->n_rows, n = hi - lo;
        AsLoc L[2];
        bool used[2] = {false, false};
        for (const AsOutCol& c : R->cols) if (c.side < 2 && !c.from_ids) used[c.side] = true;
        for (int sd = 0; sd < 2; ++sd) if (used[sd]) as_locate(*R->t[sd], R->idx[sd], lo, hi, L[sd], R->threads);
        auto* o = new AsBatchOwner();
        std::unique_ptr<AsBatchOwner> guard(o);
        o->child.resize(R->cols.size());
        for (ArrowArray& c : o->child) c.release = nullptr;
        for (size_t c = 0; c < R->cols.size(); ++c) {
            const AsOutCol& oc = R->cols[c];
            const int rc = oc.side == 2 ? as_extra_col(*R, lo, n, &o->child[c])
                           : (oc.from_ids ? as_chrom_col(*R, oc, lo, n, R->threads, &o->child[c])
                                          : as_gather_col(*R->t[oc.side], oc, L[oc.side], n, R->threads, &o->child[c]));
            if (rc != IVJ_OK) {
                R->last_error = g_err;
                for (ArrowArray& d : o->child) if (d.release) d.release(&d);
                return rc == IVJ_ENOMEM ? ENOMEM : EINVAL;
            }
        }
        o->ptrs.resize(o->child.size());
        for (size_t c = 0; c < o->child.size(); ++c) o->ptrs[c] = &o->child[c];
        *out = ArrowArray{n, 0, 0, 1, (int64_t)o->child.size(), o->top_buf, o->ptrs.data(), nullptr, as_release_batch, o};
        guard.release();
        R->cursor = hi;
        return 0;
    } catch (const std::bad_alloc&) { R->last_error = "out of memory"; return ENOMEM; }
    catch (const std::exception& e) { R->last_error = e.what(); return EINVAL; }
}
int as_stream_get_next(ArrowArrayStream* s, ArrowArray* out) {
    auto* R = static_cast<AsResult*>(s->private_data);
    std::lock_guard<std::mutex> lk(R->mu);
    return as_next_batch(R, out);
}
const char* as_stream_last_error(ArrowArrayStream* s) {
    auto* R = static_cast<AsResult*>(s->private_data);
    return R->last_error.empty() ? nullptr : R->last_error.c_str();
}
void as_stream_release(ArrowArrayStream* s) {
    delete static_cast<AsResult*>(s->private_data);
    s->private_data = nullptr;
    s->release = nullptr;
}
void as_publish(AsResult* R, ArrowArrayStream* out) {
    out->get_schema = as_stream_get_schema;
    out->get_next = as_stream_get_next;
    out->get_last_error = as_stream_last_error;
    out->release = as_stream_release;
    out->private_data = R;
}

// every column of side sd, suffixed; refuses column kinds the gather does not handle
int as_add_side_cols(AsResult& R, int sd, const char* suffix) {
    const AsTable& t = *R.t[sd];
    for (int c = 0; c < t.ncols(); ++c) {
        const AsType& ty = t.types[(size_t)c];
        if (ty.kind == AS_UNSUPPORTED)
            return fail(IVJ_EINVAL, std::string(sd == 0 ? "df1" : "df2") + ": column '" + t.name(c) + "' has Arrow format '" + (t.schema.children[c]->format ? t.schema.children[c]->format : "") +
                                        "', which the row assembly does not handle (fixed-width, bool, utf8 / binary and dictionaries of those are); project it away or gather it by the index pairs of ivj_overlap");
        R.cols.push_back(AsOutCol{sd, c, std::string(t.name(c)) + (suffix ? suffix : ""), ty});
    }
    return IVJ_OK;
}

int as_check_common(ivj_ctx* ctx, const ivj_opts* opts, ArrowArrayStream* out) {
    if (!ctx || !out) return fail(IVJ_EINVAL, "ctx or out is NULL");
    IVJ_TRY(check_opts(opts));
    return IVJ_OK;
}

}  // namespace

extern "C" {

/* gathers rows idx[0 .. n) (int64; negative = a null row) of the drained stream into a new stream of batch_rows-row batches:
 * the column gather of the joined rows on its own (needs no device) */
int ivj_arrow_take_stream(void* in_stream, const int64_t* idx, int64_t n, int64_t batch_rows, void* out_stream) try {
    if (!in_stream || !out_stream || n < 0 || (n > 0 && !idx)) return fail(IVJ_EINVAL, "take stream: bad argument");
    auto R = std::make_unique<AsResult>();
    R->t[0] = std::make_shared<AsTable>();
    IVJ_TRY(as_drain(static_cast<ArrowArrayStream*>(in_stream), "input", *R->t[0]));
    IVJ_TRY(as_add_side_cols(*R, 0, ""));
    R->own_idx[0].resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) R->own_idx[0][(size_t)i] = (idx[i] < 0 || idx[i] >= R->t[0]->n) ? -1 : (int32_t)idx[i];
    R->idx[0] = R->own_idx[0].data();
    R->n_rows = n;
    if (batch_rows > 0) R->batch_rows = batch_rows;
    as_publish(R.release(), static_cast<ArrowArrayStream*>(out_stream));
    return IVJ_OK;
} IVJ_ABI_CATCH

/* The key columns the join sees, for hosts that keep the row assembly to themselves: both streams drained, chrom encoded with
 * ONE dictionary over both sides, start / end narrowed to int32 (range-checked).  Library-owned host buffers, free with
 * ivj_arrow_keys_free.  Needs no device. */
int ivj_arrow_encode_keys(void* df1_stream, void* df2_stream, const char* const* cols1, const char* const* cols2, ivj_arrow_keys* out) try {
    if (!df1_stream || !df2_stream || !out) return fail(IVJ_EINVAL, "encode keys: NULL argument");
    std::memset(out, 0, sizeof(*out));
    AsTable t1, t2;
    // both streams are consumed whatever happens
    const int r1 = as_drain(static_cast<ArrowArrayStream*>(df1_stream), "df1", t1);
    const std::string why1 = g_err;
    const int r2 = as_drain(static_cast<ArrowArrayStream*>(df2_stream), "df2", t2);
    if (r1 != IVJ_OK) { g_err = why1; return r1; }
    if (r2 != IVJ_OK) return r2;
    AsKeys K;
    IVJ_TRY(as_make_keys(t1, t2, cols1, cols2, K, 0));
    auto dup = [](const std::vector<int32_t>& v) -> int32_t* {
        int32_t* p = (int32_t*)std::malloc(v.empty() ? 4 : v.size() * 4);
        if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * 4);
        return p;
    };
    out->n1 = t1.n; out->n2 = t2.n;
    out->contig1 = dup(K.c1); out->start1 = dup(K.s1); out->end1 = dup(K.e1);
    out->contig2 = dup(K.c2); out->start2 = dup(K.s2); out->end2 = dup(K.e2);
    out->n_contigs = (int32_t)K.dict.names.size();
    size_t bytes = 0;
    for (const std::string& s : K.dict.names) bytes += s.size();
    out->name_offsets = (int64_t*)std::malloc((K.dict.names.size() + 1) * 8);
    out->name_bytes = (char*)std::malloc(bytes ? bytes : 1);
    if (!out->contig1 || !out->start1 || !out->end1 || !out->contig2 || !out->start2 || !out->end2 || !out->name_offsets || !out->name_bytes) {
        ivj_arrow_keys_free(out);
        return fail(IVJ_ENOMEM, "encode keys: out of memory");
    }
    int64_t o = 0;
    for (size_t v = 0; v < K.dict.names.size(); ++v) {
        out->name_offsets[v] = o;
        std::memcpy(out->name_bytes + o, K.dict.names[v].data(), K.dict.names[v].size());
        o += (int64_t)K.dict.names[v].size();
    }
    out->name_offsets[K.dict.names.size()] = o;
    return IVJ_OK;
} IVJ_ABI_CATCH

void ivj_arrow_keys_free(ivj_arrow_keys* k) {
    if (!k) return;
    std::free(k->contig1); std::free(k->start1); std::free(k->end1);
    std::free(k->contig2); std::free(k->start2); std::free(k->end2);
    std::free(k->name_offsets); std::free(k->name_bytes);
    std::memset(k, 0, sizeof(*k));
}

namespace {
struct AsCall {
    std::unique_ptr<AsResult> R;
    AsKeys K;
    ivj_opts opts;
    ivj_side probe, build;
    int chrom_col[2] = {-1, -1};
};
void as_keys_to_result(AsCall& C);
int as_open(ivj_ctx* ctx, void* df1, void* df2, const char* const* cols1, const char* const* cols2, const ivj_opts* opts, void* out, AsCall& C) {
    IVJ_TRY(as_check_common(ctx, opts, static_cast<ArrowArrayStream*>(out)));
    if (!df1 || !df2) return fail(IVJ_EINVAL, "an input stream is NULL");
    C.R = std::make_unique<AsResult>();
    C.R->t[0] = std::make_shared<AsTable>();
    C.R->t[1] = std::make_shared<AsTable>();
    {
        // both streams are consumed whatever happens (as_drain releases the one it was given; the other one goes here)
        const int r1 = as_drain(static_cast<ArrowArrayStream*>(df1), "df1", *C.R->t[0]);
        if (r1 != IVJ_OK) { auto* s2 = static_cast<ArrowArrayStream*>(df2); const std::string why = g_err; if (s2->release) s2->release(s2); g_err = why; return r1; }
    }
    IVJ_TRY(as_drain(static_cast<ArrowArrayStream*>(df2), "df2", *C.R->t[1]));
    IVJ_TRY(as_make_keys(*C.R->t[0], *C.R->t[1], cols1, cols2, C.K, 0));
    C.opts = *opts;
    C.opts.n_contigs = (int32_t)C.K.dict.names.size();      // the dictionary is made here: the caller's value is not looked at
    C.probe = ivj_side{C.K.c1.data(), C.K.s1.data(), C.K.e1.data(), C.R->t[0]->n, nullptr};
    C.build = ivj_side{C.K.c2.data(), C.K.s2.data(), C.K.e2.data(), C.R->t[1]->n, nullptr};
    static const char* const dflt[3] = {"chrom", "start", "end"};
    C.chrom_col[0] = C.R->t[0]->find((cols1 ? cols1 : dflt)[0]);
    C.chrom_col[1] = C.R->t[1]->find((cols2 ? cols2 : dflt)[0]);
    return IVJ_OK;
}
// after the join (the key vectors are no longer needed as inputs): the chrom key columns of the result come from the ids
void as_keys_to_result(AsCall& C) {
    AsResult& R = *C.R;
    R.chrom_ids[0] = std::move(C.K.c1); R.chrom_ids[1] = std::move(C.K.c2);
    R.names = std::move(C.K.dict.names);
    for (AsOutCol& oc : R.cols)
        if (oc.side < 2 && oc.col == C.chrom_col[oc.side]) oc.from_ids = true;
}
}  // namespace

namespace {
// ---- the LAZY form of the one-call entry (round 5): df1 is never materialised whole ---------------------------------------------------
// /root/reference/src/lib.rs:154-214 (range_operation_lazy) registers both Arrow streams as streaming tables and the result is pulled
// batch by batch (src/scan.rs:294-357: fan-out with back-pressure, buffer 2; polars_bio/range_op_io.py:100-174).  Here: df2 is drained,
// encoded and indexed ONCE (it is the build side: it has to be whole); df1 stays a stream.  Every get_next of the RESULT stream pulls
// as many df1 batches as it needs to have a result batch: a batch is encoded with the session's chrom dictionary (names df2 does not
// have get ids beyond the dictionary: they match nothing), narrowed to int32 with the range check, SUBMITTED to a streaming probe
// session (ivj_stream_*: its H2D copy overlaps the join of the batch before and the D2H copy of the batch before that), and the
// batch that comes back -- the one submitted two turns earlier -- is assembled into record batches of batch_rows rows.  Host memory:
// df2 + three df1 batches + one batch's result, whatever the length of df1; df1 may hold more than 2^31 rows.  A df1 batch above
// max_batch_rows is submitted in slices.  Errors of a LATER batch (a coordinate beyond int32, a column that vanished) surface from
// get_next (errno + get_last_error), as the reference's do at collect time.
struct AsLazy {
    ivj_ctx* ctx = nullptr;
    ivj_stream* st = nullptr;
    int op = 0, k = 1, with_distance = 0;
    ArrowArrayStream in{};                               // df1: moved in, released with the result stream (or at its end)
    ArrowSchema schema1{};
    std::vector<AsType> types1;
    int key1[3] = {-1, -1, -1};
    std::shared_ptr<AsTable> t2;
    std::vector<int32_t> c2;                             // chrom ids of df2
    AsDict dict;
    AsResult proto;                                      // the result's columns
    int64_t max_rows = 0, coalesce_rows = 1, batch_rows = 1 << 20, limit = -1, rows_out = 0;
    ArrowArray carry{};                                  // a pulled df1 batch that did not fit the group before it
    struct Pending { std::shared_ptr<AsTable> tb; std::shared_ptr<std::vector<int32_t>> cids; int64_t off, n; };
    std::deque<Pending> pending;                         // submitted slices whose results are still in the session (delivery order)
    std::deque<ArrowArray> ready;
    std::shared_ptr<AsTable> cur;                        // the df1 batch being sliced, its keys
    std::shared_ptr<std::vector<int32_t>> cur_c;
    std::vector<int32_t> cur_s, cur_e;
    int64_t cur_off = 0;
    bool in_done = false, finished = false;
    int sticky_errno = 0;                 // first get_next failure, reported again by every later call
    double t_pull = 0, t_turn = 0, t_asm = 0;            // IVJ_DEBUG_TIMES: seconds in df1 pull + key encoding / session turns / batch assembly
    std::string last_error;
    std::mutex mu;
    AsLazy() { in.release = nullptr; schema1.release = nullptr; carry.release = nullptr; }
    ~AsLazy() {
        for (ArrowArray& a : ready) if (a.release) a.release(&a);
        if (st) ivj_stream_close(st);
        if (carry.release) carry.release(&carry);
        if (in.release) in.release(&in);
        if (schema1.release) schema1.release(&schema1);
    }
};

// The next GROUP of df1 batches as a table view + its keys; L.cur stays empty at the end of the stream.  Batches below the slice size
// are coalesced (round 5): the library pulls until the group holds coalesce_rows rows (min(max_batch_rows, 2 Mi)) or the stream ends --
// a df1 that arrives in 16 batches of 625 k rows took 16 turns with their per-turn fixed costs and 2-MB result columns (0.094 s for the
// 10 M x 1 M call against 0.060 s for the same rows in one batch); a batch that would take the group beyond twice that size (or 2^31 - 1
// rows) waits in L.carry for the next group.  Host memory: three groups instead of three batches.  A call with a row limit does not
// coalesce: it may need only the first batch.
int lazy_pull(AsLazy& L) {
    L.cur.reset(); L.cur_c.reset(); L.cur_off = 0;
    auto tb = std::make_shared<AsTable>();
    tb->schema = L.schema1; tb->schema.release = nullptr;                       // a view: the session owns the schema
    tb->types = L.types1;
    tb->start = {0};
    while (tb->n < L.coalesce_rows) {
        ArrowArray a{};
        a.release = nullptr;
        if (L.carry.release) { a = L.carry; L.carry.release = nullptr; }
        else {
            if (L.in_done) break;
            if (L.in.get_next(&L.in, &a) != 0) {
                const char* m = L.in.get_last_error ? L.in.get_last_error(&L.in) : nullptr;
                return fail(IVJ_EINVAL, std::string("df1: get_next failed") + (m ? std::string(": ") + m : std::string()));
            }
            if (!a.release) { L.in_done = true; if (L.in.release) L.in.release(&L.in); break; }
            const int rc = as_check_batch(L.schema1, L.types1, a, "df1");
            if (rc != IVJ_OK) { a.release(&a); return rc; }
            if (a.length == 0) { a.release(&a); continue; }
            if (a.length > (int64_t)INT32_MAX) { a.release(&a); return fail(IVJ_EINVAL, "df1: a batch of more than 2^31 - 1 rows"); }
        }
        if (tb->n > 0 && (tb->n + a.length > 2 * L.coalesce_rows || tb->n + a.length > (int64_t)INT32_MAX)) { L.carry = a; break; }
        tb->batches.push_back(a);
        tb->n += a.length;
        tb->start.push_back(tb->n);
    }
    if (tb->n == 0) return IVJ_OK;
    auto ids = std::make_shared<std::vector<int32_t>>((size_t)tb->n);
    L.cur_s.resize((size_t)tb->n); L.cur_e.resize((size_t)tb->n);
    IVJ_TRY(as_encode_chrom(*tb, L.key1[0], "df1", L.dict, ids->data(), 0));
    IVJ_TRY(as_narrow_coord(*tb, L.key1[1], "df1", L.cur_s.data(), 0));
    IVJ_TRY(as_narrow_coord(*tb, L.key1[2], "df1", L.cur_e.data(), 0));
    L.cur = tb; L.cur_c = ids;
    return IVJ_OK;
}

// results of one delivered slice -> record batches in L.ready
int lazy_assemble(AsLazy& L, const AsLazy::Pending& P, const ivj_stream_result& d) {
    AsResult B;
    B.t[0] = P.tb; B.t[1] = L.t2;
    B.cols = L.proto.cols;
    B.cid[0] = P.cids->data(); B.cid_n[0] = (int64_t)P.cids->size();
    B.cid[1] = L.c2.data(); B.cid_n[1] = (int64_t)L.c2.size();
    B.names_ref = &L.dict.names;
    B.batch_rows = L.batch_rows;
    const int32_t off = (int32_t)P.off;
    if (L.op == IVJ_STREAM_OVERLAP) {
        B.own_idx[0].resize((size_t)d.n);
        for (int64_t i = 0; i < d.n; ++i) B.own_idx[0][(size_t)i] = d.probe_idx[i] + off;
        B.idx[0] = B.own_idx[0].data();
        B.idx[1] = d.build_idx;                                                // the session's pinned slot: valid until the next turn -- the batches are made now
        B.n_rows = d.n;
    } else if (L.op == IVJ_STREAM_COUNT) {
        B.own_idx[0].resize((size_t)d.n_probe);
        for (int64_t i = 0; i < d.n_probe; ++i) B.own_idx[0][(size_t)i] = (int32_t)i + off;
        B.idx[0] = B.own_idx[0].data();
        B.extra.assign(d.counts, d.counts + d.n_probe);
        B.n_rows = d.n_probe;
    } else {
        for (int64_t i = 0; i < d.n_probe; ++i) {
            const int32_t f = d.n_found[i];
            for (int32_t j = 0; j < (f > 0 ? f : 1); ++j) {
                const bool none = f <= 0;
                B.own_idx[0].push_back((int32_t)i + off);
                B.own_idx[1].push_back(none ? -1 : d.build_idx[i * L.k + j]);
                B.extra.push_back(none ? 0 : d.dist[i * L.k + j]);
                B.extra_null.push_back(none ? 1 : 0);
            }
        }
        B.idx[0] = B.own_idx[0].data(); B.idx[1] = B.own_idx[1].data();
        B.n_rows = (int64_t)B.own_idx[0].size();
    }
    if (L.limit >= 0 && L.rows_out + B.n_rows > L.limit) B.n_rows = L.limit - L.rows_out;
    for (;;) {
        ArrowArray a{};
        const int rc = as_next_batch(&B, &a);
        if (rc != 0) return fail(rc == ENOMEM ? IVJ_ENOMEM : IVJ_EINVAL, B.last_error.empty() ? std::string("result batch assembly failed") : B.last_error);
        if (!a.release) break;
        L.rows_out += a.length;
        L.ready.push_back(a);
    }
    return IVJ_OK;
}

int lazy_get_next(ArrowArrayStream* s, ArrowArray* out) {
    auto* L = static_cast<AsLazy*>(s->private_data);
    std::lock_guard<std::mutex> lk(L->mu);
    // a failure is STICKY: every later get_next reports the same errno (a consumer that retries must not read a truncated result as complete)
    auto bad = [&](int rc) { L->last_error = g_err; L->finished = true; L->sticky_errno = rc == IVJ_ENOMEM ? ENOMEM : (rc == IVJ_EINVAL ? EINVAL : EIO); return L->sticky_errno; };
    if (L->sticky_errno) return L->sticky_errno;
    try {
        for (;;) {
            if (!L->ready.empty()) { *out = L->ready.front(); L->ready.pop_front(); return 0; }
            if (L->finished) { std::memset(out, 0, sizeof(*out)); out->release = nullptr; return 0; }
            if (L->limit >= 0 && L->rows_out >= L->limit) {                    // enough rows: the rest of df1 is never pulled
                L->finished = true;
                if (L->in.release) L->in.release(&L->in);
                continue;
            }
            auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
            double t0 = now();
            if (!L->cur || L->cur_off >= L->cur->n) { const int rc = lazy_pull(*L); if (rc != IVJ_OK) return bad(rc); }
            L->t_pull += now() - t0; t0 = now();
            ivj_stream_result done;
            std::memset(&done, 0, sizeof(done));
            done.batch = -1;
            int rc;
            if (L->cur && L->cur_off < L->cur->n) {
                const int64_t n = std::min<int64_t>(L->max_rows, L->cur->n - L->cur_off);
                const ivj_side side{L->cur_c->data() + L->cur_off, L->cur_s.data() + L->cur_off, L->cur_e.data() + L->cur_off, n, nullptr};
                L->pending.push_back(AsLazy::Pending{L->cur, L->cur_c, L->cur_off, n});
                L->cur_off += n;
                rc = ivj_stream_submit(L->st, &side, &done);
            } else {
                rc = ivj_stream_flush(L->st, &done);
                if (rc == IVJ_OK && done.batch < 0) {
                    L->finished = true;
                    if (std::getenv("IVJ_DEBUG_TIMES")) std::fprintf(stderr, "[ivj] lazy arrow stream: pull + encode %.4f s, session turns %.4f s, assembly %.4f s, %lld rows\n", L->t_pull, L->t_turn, L->t_asm, (long long)L->rows_out);
                    continue;
                }
            }
            L->t_turn += now() - t0; t0 = now();
            if (rc != IVJ_OK) return bad(rc);
            if (done.batch >= 0) {
                if (L->pending.empty()) { g_err = "lazy stream: a result without a pending batch (internal)"; return bad(IVJ_ESTATE); }
                const AsLazy::Pending P = L->pending.front();
                L->pending.pop_front();
                rc = lazy_assemble(*L, P, done);
                L->t_asm += now() - t0;
                if (rc != IVJ_OK) return bad(rc);
            }
        }
    } catch (const std::bad_alloc&) { L->last_error = "out of memory"; L->finished = true; L->sticky_errno = ENOMEM; return ENOMEM; }
    catch (const std::exception& e) { L->last_error = e.what(); L->finished = true; L->sticky_errno = EINVAL; return EINVAL; }
}
int lazy_get_schema(ArrowArrayStream* s, ArrowSchema* out) {
    auto* L = static_cast<AsLazy*>(s->private_data);
    try { return as_result_schema(L->proto, out) == IVJ_OK ? 0 : EINVAL; }
    catch (const std::exception& e) { L->last_error = e.what(); return ENOMEM; }
}
const char* lazy_last_error(ArrowArrayStream* s) {
    auto* L = static_cast<AsLazy*>(s->private_data);
    return L->last_error.empty() ? nullptr : L->last_error.c_str();
}
void lazy_release(ArrowArrayStream* s) {
    delete static_cast<AsLazy*>(s->private_data);
    s->private_data = nullptr;
    s->release = nullptr;
}

// df2 drained / encoded / indexed, df1's schema read, the result's columns laid out; on failure both input streams are released
int lazy_open(ivj_ctx* ctx, void* df1, void* df2, const char* const* cols1, const char* const* cols2, const ivj_opts* opts, int op,
              const char* suffix1, const char* suffix2, int with_distance, int64_t batch_rows, int64_t max_batch_rows, int64_t limit, void* out) {
    auto* s1 = static_cast<ArrowArrayStream*>(df1);
    auto* s2 = static_cast<ArrowArrayStream*>(df2);
    auto L = std::make_unique<AsLazy>();
    auto fail_both = [&](int rc) {                        // (df1 may already live in L: its destructor releases it)
        const std::string why = g_err;
        if (s1 && s1->release) s1->release(s1);
        if (s2 && s2->release) s2->release(s2);
        g_err = why;
        return rc;
    };
    {
        const int rc = as_check_common(ctx, opts, static_cast<ArrowArrayStream*>(out));
        if (rc != IVJ_OK) return fail_both(rc);
    }
    if (!s1 || !s2 || !s1->get_schema || !s1->get_next) return fail_both(fail(IVJ_EINVAL, "an input stream is NULL or not an ArrowArrayStream"));
    L->ctx = ctx; L->op = op; L->with_distance = with_distance; L->limit = limit;
    if (batch_rows > 0) L->batch_rows = batch_rows;
    L->max_rows = max_batch_rows > 0 ? max_batch_rows : (4ll << 20);
    if (L->max_rows > 0x7fff0000ll) L->max_rows = 0x7fff0000ll;
    L->coalesce_rows = limit >= 0 ? 1 : std::min<int64_t>(L->max_rows, 2ll << 20);   // (a call with a row limit pulls df1 one batch at a time: it may need very little of it)
    L->in = *s1;                                          // moved: the caller's struct is marked released (Arrow C stream move semantics)
    s1->release = nullptr;
    s1 = nullptr;
    if (L->in.get_schema(&L->in, &L->schema1) != 0) {
        const char* m = L->in.get_last_error ? L->in.get_last_error(&L->in) : nullptr;
        return fail_both(fail(IVJ_EINVAL, std::string("df1: get_schema failed") + (m ? std::string(": ") + m : std::string())));
    }
    if (!L->schema1.format || std::strcmp(L->schema1.format, "+s") != 0) return fail_both(fail(IVJ_EINVAL, "df1: the stream's schema is not a struct (record batches expected)"));
    L->types1.resize((size_t)L->schema1.n_children);
    for (int64_t c = 0; c < L->schema1.n_children; ++c) L->types1[(size_t)c] = as_type_of(L->schema1.children[c]);
    L->t2 = std::make_shared<AsTable>();
    {
        const int rc = as_drain(s2, "df2", *L->t2);       // (releases df2 whatever happens)
        s2 = nullptr;
        if (rc != IVJ_OK) return fail_both(rc);
    }
    static const char* const dflt[3] = {"chrom", "start", "end"};
    const char* const* n1 = cols1 ? cols1 : dflt;
    const char* const* n2 = cols2 ? cols2 : dflt;
    int i2[3];
    auto find1 = [&](const char* nm) { for (int64_t c = 0; c < L->schema1.n_children; ++c) if (L->schema1.children[c]->name && !std::strcmp(L->schema1.children[c]->name, nm)) return (int)c; return -1; };
    for (int q = 0; q < 3; ++q) {
        if (!n1[q] || !n2[q]) return fail_both(fail(IVJ_EINVAL, "a key column name is NULL"));
        L->key1[q] = find1(n1[q]); i2[q] = L->t2->find(n2[q]);
        if (L->key1[q] < 0) return fail_both(fail(IVJ_EINVAL, std::string("df1: column '") + n1[q] + "' not found"));
        if (i2[q] < 0) return fail_both(fail(IVJ_EINVAL, std::string("df2: column '") + n2[q] + "' not found"));
    }
    // df2's keys: the dictionary starts with ITS names, so opts.n_contigs = names so far covers every build row
    std::vector<int32_t> s2v((size_t)L->t2->n), e2v((size_t)L->t2->n);
    L->c2.resize((size_t)L->t2->n);
    int rc = as_encode_chrom(*L->t2, i2[0], "df2", L->dict, L->c2.data(), 0);
    if (rc == IVJ_OK) rc = as_narrow_coord(*L->t2, i2[1], "df2", s2v.data(), 0);
    if (rc == IVJ_OK) rc = as_narrow_coord(*L->t2, i2[2], "df2", e2v.data(), 0);
    if (rc != IVJ_OK) return fail_both(rc);
    ivj_opts o = *opts;
    o.n_contigs = (int32_t)L->dict.names.size();
    if (op == IVJ_STREAM_NEAREST) { if (o.nearest_k < 1) o.nearest_k = 1; if (o.nearest_k > 1024) return fail_both(fail(IVJ_EINVAL, "nearest_k > 1024")); }
    L->k = o.nearest_k < 1 ? 1 : o.nearest_k;
    // the result's columns: df1's from a one-batch view per pulled batch, df2's from the drained table
    {
        auto v1 = std::make_shared<AsTable>();
        v1->schema = L->schema1; v1->schema.release = nullptr; v1->types = L->types1;
        L->proto.t[0] = v1; L->proto.t[1] = L->t2;
        rc = as_add_side_cols(L->proto, 0, suffix1 ? suffix1 : (op == IVJ_STREAM_COUNT ? "" : "_1"));
        if (rc == IVJ_OK && op != IVJ_STREAM_COUNT) rc = as_add_side_cols(L->proto, 1, suffix2 ? suffix2 : "_2");
        if (rc != IVJ_OK) return fail_both(rc);
        if (op == IVJ_STREAM_COUNT) L->proto.cols.push_back(AsOutCol{2, 0, "count", AsType()});
        if (op == IVJ_STREAM_NEAREST && with_distance) L->proto.cols.push_back(AsOutCol{2, 0, "distance", AsType()});
        for (AsOutCol& oc : L->proto.cols)
            if ((oc.side == 0 && oc.col == L->key1[0]) || (oc.side == 1 && oc.col == i2[0])) oc.from_ids = true;
        L->proto.t[0].reset(); L->proto.t[1].reset();
    }
    const ivj_side build{L->c2.data(), s2v.data(), e2v.data(), L->t2->n, nullptr};
    rc = ivj_stream_open(ctx, &build, &o, op, L->max_rows, &L->st);
    if (rc != IVJ_OK) return fail_both(rc);
    auto* os = static_cast<ArrowArrayStream*>(out);
    os->get_schema = lazy_get_schema; os->get_next = lazy_get_next; os->get_last_error = lazy_last_error; os->release = lazy_release;
    os->private_data = L.release();
    return IVJ_OK;
}
}  // namespace

int ivj_overlap_arrow_stream(ivj_ctx* ctx, void* df1_stream, void* df2_stream, const char* const* cols1, const char* const* cols2,
                             const ivj_opts* opts, const char* suffix1, const char* suffix2, int64_t batch_rows, int64_t limit, void* out_stream) try {
    AsCall C;
    IVJ_TRY(as_open(ctx, df1_stream, df2_stream, cols1, cols2, opts, out_stream, C));
    IVJ_TRY(as_add_side_cols(*C.R, 0, suffix1 ? suffix1 : "_1"));
    IVJ_TRY(as_add_side_cols(*C.R, 1, suffix2 ? suffix2 : "_2"));
    IVJ_TRY(ivj_overlap(ctx, &C.probe, &C.build, &C.opts, &C.R->pairs));
    C.R->idx[0] = C.R->pairs.probe_idx; C.R->idx[1] = C.R->pairs.build_idx;
    C.R->n_rows = (limit >= 0 && limit < C.R->pairs.n_pairs) ? limit : C.R->pairs.n_pairs;
    if (batch_rows > 0) C.R->batch_rows = batch_rows;
    as_keys_to_result(C);
    as_publish(C.R.release(), static_cast<ArrowArrayStream*>(out_stream));
    return IVJ_OK;
} IVJ_ABI_CATCH

int ivj_count_overlaps_arrow_stream(ivj_ctx* ctx, void* df1_stream, void* df2_stream, const char* const* cols1, const char* const* cols2,
                                    const ivj_opts* opts, const char* suffix1, int64_t batch_rows, int64_t limit, void* out_stream) try {
    AsCall C;
    IVJ_TRY(as_open(ctx, df1_stream, df2_stream, cols1, cols2, opts, out_stream, C));
    IVJ_TRY(as_add_side_cols(*C.R, 0, suffix1 ? suffix1 : ""));
    C.R->extra.resize((size_t)C.probe.n);
    IVJ_TRY(ivj_count_overlaps(ctx, &C.probe, &C.build, &C.opts, C.R->extra.data()));
    C.R->cols.push_back(AsOutCol{2, 0, "count", AsType()});
    C.R->n_rows = (limit >= 0 && limit < C.probe.n) ? limit : C.probe.n;
    if (batch_rows > 0) C.R->batch_rows = batch_rows;
    as_keys_to_result(C);
    as_publish(C.R.release(), static_cast<ArrowArrayStream*>(out_stream));
    return IVJ_OK;
} IVJ_ABI_CATCH

int ivj_nearest_arrow_stream(ivj_ctx* ctx, void* df1_stream, void* df2_stream, const char* const* cols1, const char* const* cols2,
                             const ivj_opts* opts, const char* suffix1, const char* suffix2, int32_t with_distance, int64_t batch_rows,
                             int64_t limit, void* out_stream) try {
    AsCall C;
    IVJ_TRY(as_open(ctx, df1_stream, df2_stream, cols1, cols2, opts, out_stream, C));
    if (C.opts.nearest_k < 1) C.opts.nearest_k = 1;
    if (C.opts.nearest_k > 1024) return fail(IVJ_EINVAL, "nearest_k > 1024");
    IVJ_TRY(as_add_side_cols(*C.R, 0, suffix1 ? suffix1 : "_1"));
    IVJ_TRY(as_add_side_cols(*C.R, 1, suffix2 ? suffix2 : "_2"));
    const int64_t n = C.probe.n, k = C.opts.nearest_k;
    std::vector<int32_t> idx((size_t)(n * k)), nf((size_t)n);
    std::vector<int64_t> dist((size_t)(n * k));
    IVJ_TRY(ivj_nearest(ctx, &C.probe, &C.build, &C.opts, idx.data(), dist.data(), nf.data()));
    // one result row per (probe row, found neighbour); a probe row without any keeps ONE row with null df2 columns and a null
    // distance (the reference leaves that case unpinned; this is the front door's rule, range_op.py::_assemble_nearest)
    AsResult& R = *C.R;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t f = nf[(size_t)i];
        for (int32_t j = 0; j < (f > 0 ? f : 1); ++j) {
            const bool none = f <= 0;
            R.own_idx[0].push_back((int32_t)i);
            R.own_idx[1].push_back(none ? -1 : idx[(size_t)(i * k + j)]);
            R.extra.push_back(none ? 0 : dist[(size_t)(i * k + j)]);
            R.extra_null.push_back(none ? 1 : 0);
        }
    }
    R.idx[0] = R.own_idx[0].data(); R.idx[1] = R.own_idx[1].data();
    if (with_distance) R.cols.push_back(AsOutCol{2, 0, "distance", AsType()});
    const int64_t rows = (int64_t)R.own_idx[0].size();
    R.n_rows = (limit >= 0 && limit < rows) ? limit : rows;
    if (batch_rows > 0) R.batch_rows = batch_rows;
    as_keys_to_result(C);
    as_publish(C.R.release(), static_cast<ArrowArrayStream*>(out_stream));
    return IVJ_OK;
} IVJ_ABI_CATCH

int ivj_overlap_arrow_stream_lazy(ivj_ctx* ctx, void* df1_stream, void* df2_stream, const char* const* cols1, const char* const* cols2,
                                  const ivj_opts* opts, const char* suffix1, const char* suffix2, int64_t batch_rows, int64_t max_batch_rows,
                                  int64_t limit, void* out_stream) try {
    return lazy_open(ctx, df1_stream, df2_stream, cols1, cols2, opts, IVJ_STREAM_OVERLAP, suffix1, suffix2, 0, batch_rows, max_batch_rows, limit, out_stream);
} IVJ_ABI_CATCH

int ivj_count_overlaps_arrow_stream_lazy(ivj_ctx* ctx, void* df1_stream, void* df2_stream, const char* const* cols1, const char* const* cols2,
                                         const ivj_opts* opts, const char* suffix1, int64_t batch_rows, int64_t max_batch_rows, int64_t limit,
                                         void* out_stream) try {
    return lazy_open(ctx, df1_stream, df2_stream, cols1, cols2, opts, IVJ_STREAM_COUNT, suffix1, nullptr, 0, batch_rows, max_batch_rows, limit, out_stream);
} IVJ_ABI_CATCH

int ivj_nearest_arrow_stream_lazy(ivj_ctx* ctx, void* df1_stream, void* df2_stream, const char* const* cols1, const char* const* cols2,
                                  const ivj_opts* opts, const char* suffix1, const char* suffix2, int32_t with_distance, int64_t batch_rows,
                                  int64_t max_batch_rows, int64_t limit, void* out_stream) try {
    return lazy_open(ctx, df1_stream, df2_stream, cols1, cols2, opts, IVJ_STREAM_NEAREST, suffix1, suffix2, with_distance, batch_rows, max_batch_rows, limit, out_stream);
} IVJ_ABI_CATCH

}  // extern "C"
