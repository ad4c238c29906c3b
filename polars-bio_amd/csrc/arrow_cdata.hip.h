// arrow_cdata.hip.h -- Arrow C Data Interface: import of one side, export of a materialised result (included inside ivjoin.hip's extern "C" block)
// Part of the single translation unit ivjoin.hip (included there, in this order); not a stand-alone header.
#pragma once

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

int ivj_side_from_arrow(const void* array, const void* schema, ivj_side* out) try {
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
} IVJ_ABI_CATCH

int ivj_rows_export_arrow(ivj_rows* rows, void* out_array, void* out_schema) try {
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
} IVJ_ABI_CATCH
