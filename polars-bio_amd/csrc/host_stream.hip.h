// host_stream.hip.h -- the streaming probe session (host <-> HBM helpers: host_mem.hip.h)
// Part of the single translation unit ivjoin.hip (included there, in this order); not a stand-alone header.
#pragma once

// ===================================================================== streaming probe session
// The reference streams the probe side through its executor batch by batch (range_op_io.py:100-174 feeds an Arrow C stream,
// src/lib.rs:154-214 / src/scan.rs:294-357 fan the batches out with back-pressure) and yields result batches lazily.  Here:
// the build side is indexed once and stays in HBM; probe batches are SUBMITTED one at a time and every submit overlaps
//     H2D of the batch just handed over   (copy stream, out of a pinned staging slot)
//     the join of the previous batch      (compute stream = the context's stream)
//     D2H of the batch before that        (copy-back stream, into a pinned result slot)
// and hands out the finished results of the batch submitted two calls earlier.  Three slots rotate.
struct ivj_stream {
    ivj_ctx* ctx = nullptr;
    ivj_index* ix = nullptr;
    ivj_opts opts{};
    int op = 0;                              // IVJ_STREAM_OVERLAP / COUNT / NEAREST
    int k = 1;
    int64_t max_rows = 0;
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    struct Slot {
        int64_t id = -1, n = 0, n_out = 0, off2 = 0;        // off2: overlap, device offset (elements) of the build_idx column
        int32_t* h_in = nullptr;             // pinned: contig | start | end, max_rows each
        int32_t* d_in = nullptr;
        size_t in_cap = 0;                   // bytes of h_in / d_in
        char* d_out = nullptr; size_t d_out_cap = 0;
        char* h_out = nullptr; size_t h_out_cap = 0;     // pinned
        hipEvent_t ev_h2d = nullptr, ev_join = nullptr, ev_d2h = nullptr;
        int state = 0;                       // 0 free, 1 H2D enqueued, 2 D2H enqueued
    } slot[3];
    int64_t submitted = 0, joined = 0, delivered = 0;
    int64_t pair_hint = 0;                   // pairs of the last overlap batch (capacity guess of the fused pass)
};

namespace {

int stream_grow(char** p, size_t* cap, size_t need, bool pinned) {
    if (need <= *cap) return IVJ_OK;
    if (*p) { if (pinned) (void)hipHostFree(*p); else (void)hipFree(*p); *p = nullptr; *cap = 0; }
    const size_t want = align_up(need + need / 4, 1 << 20);
    hipError_t e = pinned ? hipHostMalloc((void**)p, want, hipHostMallocDefault) : hipMalloc((void**)p, want);
    if (e != hipSuccess) return fail(IVJ_ENOMEM, std::string(pinned ? "hipHostMalloc" : "hipMalloc") + "(stream result): " + hipGetErrorString(e));
    *cap = want;
    return IVJ_OK;
}

// join of the batch in slot s on the compute stream (blocks the host until its totals are known), then its D2H
int stream_join_slot(ivj_stream* st, int s) {
    ivj_ctx* ctx = st->ctx;
    ivj_stream::Slot& S = st->slot[s];
    HIP_TRY(hipStreamWaitEvent(ctx->stream, S.ev_h2d, 0));
    const size_t col = align_up((size_t)st->max_rows * 4);
    ivj_side side{S.d_in, (int32_t*)((char*)S.d_in + col), (int32_t*)((char*)S.d_in + 2 * col), S.n, nullptr};
    const int64_t n = S.n;
    size_t out_bytes = 0;
    if (n == 0) { S.n_out = 0; }
    else if (st->op == IVJ_STREAM_OVERLAP) {
        int64_t total = 0;
        int rc = IVJ_ECAPACITY;
        const int64_t guess = st->pair_hint > 0 ? st->pair_hint + st->pair_hint / 4 + 4096 : 0;
        if (guess > 0) {
            IVJ_TRY(stream_grow(&S.d_out, &S.d_out_cap, (size_t)guess * 8, false));
            rc = overlap_fused(ctx, st->ix, &side, &st->opts, (int32_t*)S.d_out, (int32_t*)S.d_out + guess, guess, &total);
            if (rc != IVJ_OK && rc != IVJ_ECAPACITY) return rc;
            S.off2 = guess;
        }
        if (rc == IVJ_ECAPACITY) {                                             // first batch, or the guess was too small: count -> fill
            IVJ_TRY(overlap_count(ctx, st->ix, &side, &st->opts, &total));
            IVJ_TRY(stream_grow(&S.d_out, &S.d_out_cap, (size_t)(total > 0 ? total : 1) * 8, false));
            IVJ_TRY(overlap_fill(ctx, st->ix, &side, &st->opts, (int32_t*)S.d_out, (int32_t*)S.d_out + total, total));
            S.off2 = total;
        }
        st->pair_hint = total;
        S.n_out = total;
        out_bytes = (size_t)total * 8;
    } else if (st->op == IVJ_STREAM_COUNT) {
        IVJ_TRY(stream_grow(&S.d_out, &S.d_out_cap, (size_t)n * 8, false));
        IVJ_TRY(count_overlaps_dev(ctx, st->ix, &side, &st->opts, (int64_t*)S.d_out));
        S.n_out = n;
        out_bytes = (size_t)n * 8;
    } else {
        const size_t slots = (size_t)n * (size_t)st->k;
        const size_t o_dist = align_up(slots * 4, 8), o_nf = o_dist + slots * 8;
        IVJ_TRY(stream_grow(&S.d_out, &S.d_out_cap, o_nf + (size_t)n * 4, false));
        IVJ_TRY(nearest_dev(ctx, st->ix, &side, &st->opts, (int32_t*)S.d_out, (int64_t*)(S.d_out + o_dist), (int32_t*)(S.d_out + o_nf)));
        S.n_out = n;
        out_bytes = o_nf + (size_t)n * 4;
    }
    HIP_TRY(hipEventRecord(S.ev_join, ctx->stream));
    HIP_TRY(hipStreamWaitEvent(st->s_d2h, S.ev_join, 0));
    if (out_bytes) {
        IVJ_TRY(stream_grow(&S.h_out, &S.h_out_cap, out_bytes, true));
        if (st->op == IVJ_STREAM_OVERLAP) {                                    // the two columns packed back to back on the host
            HIP_TRY(hipMemcpyAsync(S.h_out, S.d_out, out_bytes / 2, hipMemcpyDeviceToHost, st->s_d2h));
            HIP_TRY(hipMemcpyAsync(S.h_out + out_bytes / 2, (int32_t*)S.d_out + S.off2, out_bytes / 2, hipMemcpyDeviceToHost, st->s_d2h));
        } else HIP_TRY(hipMemcpyAsync(S.h_out, S.d_out, out_bytes, hipMemcpyDeviceToHost, st->s_d2h));
    }
    HIP_TRY(hipEventRecord(S.ev_d2h, st->s_d2h));
    S.state = 2;
    return IVJ_OK;
}

void stream_fill_result(ivj_stream* st, int s, ivj_stream_result* out) {
    ivj_stream::Slot& S = st->slot[s];
    std::memset(out, 0, sizeof(*out));
    out->batch = S.id; out->n_probe = S.n; out->n = S.n_out;
    if (st->op == IVJ_STREAM_OVERLAP) { out->probe_idx = (int32_t*)S.h_out; out->build_idx = (int32_t*)S.h_out + S.n_out; }
    else if (st->op == IVJ_STREAM_COUNT) out->counts = (int64_t*)S.h_out;
    else {
        const size_t slots = (size_t)S.n * (size_t)st->k;
        const size_t o_dist = align_up(slots * 4, 8), o_nf = o_dist + slots * 8;
        out->build_idx = (int32_t*)S.h_out; out->dist = (int64_t*)(S.h_out + o_dist); out->n_found = (int32_t*)(S.h_out + o_nf);
    }
}

// one turn of the pipeline: (optionally) take a new batch, join the previous one, deliver the one before
int stream_turn(ivj_stream* st, const ivj_side* batch, ivj_stream_result* done) {
    ivj_ctx* ctx = st->ctx;
    DeviceGuard g(ctx->device);
    if (done) { std::memset(done, 0, sizeof(*done)); done->batch = -1; }
    if (batch) {
        const int s = (int)(st->submitted % 3);
        ivj_stream::Slot& S = st->slot[s];
        if (S.state != 0) return fail(IVJ_ESTATE, "stream slot still in flight (internal)");
        const size_t col = align_up((size_t)st->max_rows * 4);
        const size_t nb = (size_t)batch->n * 4;
        if (batch->n > 0) {
            host_copy_parallel((char*)S.h_in, batch->contig, nb);
            host_copy_parallel((char*)S.h_in + col, batch->start, nb);
            host_copy_parallel((char*)S.h_in + 2 * col, batch->end, nb);
            for (int c = 0; c < 3; ++c)
                HIP_TRY(hipMemcpyAsync((char*)S.d_in + c * col, (char*)S.h_in + c * col, nb, hipMemcpyHostToDevice, st->s_h2d));
        }
        HIP_TRY(hipEventRecord(S.ev_h2d, st->s_h2d));
        S.id = st->submitted; S.n = batch->n; S.n_out = 0; S.state = 1;
        ++st->submitted;
    }
    // join the oldest batch whose columns are on their way (the host blocks here while the copy engines work)
    if (st->joined < st->submitted - (batch ? 1 : 0)) {
        IVJ_TRY(stream_join_slot(st, (int)(st->joined % 3)));
        ++st->joined;
    }
    // deliver the oldest finished batch; with a new batch in hand the join just enqueued stays in flight behind it
    if (done && st->delivered < st->joined - (batch ? 1 : 0)) {
        const int s = (int)(st->delivered % 3);
        HIP_TRY(hipEventSynchronize(st->slot[s].ev_d2h));
        stream_fill_result(st, s, done);
        st->slot[s].state = 0;
        ++st->delivered;
    }
    return IVJ_OK;
}

}  // namespace
