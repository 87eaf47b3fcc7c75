// dst_stream.cpp — stream mode behind the C ABI: batches of streamed records against the loaded set of
// slot 0 (stream(), src/lib.rs:269-365; the batches are stream_fasta()'s, src/fastaio.rs:215-286).
//
// A ring of `depth` slots, each with its own page-locked input buffer, device byte buffer, packed planes,
// device result buffer and page-locked result buffer, and three HIP streams:
//
//     copy-in   H2D of batch k+1                       |  while
//     compute   pack + pair kernel of batch k          |  all three
//     copy-out  D2H of batch k-1's results             |  run
//
// The caller encodes straight into the page-locked buffer (dst_stream_acquire), submits, and collects
// results strictly in submission order.  The byte staging of a slot is reused by the next batch that gets
// the slot; nothing is kept for the life of the context.
#include <algorithm>
#include <cstdio>
#include <new>

#include "dst_ctx.h"

using namespace dst;

struct dst_stream {
    dst_ctx *ctx = nullptr;
    int measure = 0, out_kind = 0, wire = DST_WIRE_CODES;
    size_t max_records = 0, len = 0, pitch = 0;
    size_t out_bytes_per_record = 0;
    struct Slot {
        uint8_t *h_in = nullptr, *d_in = nullptr;  // max_records x pitch codes, then max_records x 4 base counts
        void *d_out = nullptr, *h_out = nullptr;
        unsigned long long *d_bad = nullptr, *h_bad = nullptr;
        DeviceSet set;
        hipEvent_t h2d = nullptr, computed = nullptr, landed = nullptr;
        size_t n = 0;
        int state = 0;  // 0 free, 1 acquired, 2 submitted, 3 collected (results still readable)
    };
    std::vector<Slot> slots;
    hipStream_t s_in = nullptr, s_compute = nullptr, s_out = nullptr;
    size_t next_acquire = 0, next_collect = 0, in_flight = 0;
    int acquired = -1;
};

namespace {

size_t counts_offset(const dst_stream *s) { return s->max_records * s->pitch; }

void destroy(dst_stream *s)
{
    if (!s)
        return;
    (void)hipSetDevice(s->ctx->device);
    for (hipStream_t st : {s->s_in, s->s_compute, s->s_out})
        if (st)
            (void)hipStreamSynchronize(st);
    for (auto &sl : s->slots) {
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.h_bad) (void)hipHostFree(sl.h_bad);
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.d_bad) (void)hipFree(sl.d_bad);
        free_set(sl.set);
        for (hipEvent_t e : {sl.h2d, sl.computed, sl.landed})
            if (e)
                (void)hipEventDestroy(e);
    }
    for (hipStream_t st : {s->s_in, s->s_compute, s->s_out})
        if (st)
            (void)hipStreamDestroy(st);
    delete s;
}

}  // namespace

extern "C" {

int dst_stream_open(dst_ctx *ctx, int measure, int out_kind, size_t max_records, int depth, dst_stream **out)
{
    return dst_stream_open_wire(ctx, measure, out_kind, max_records, depth, DST_WIRE_CODES, out);
}

int dst_stream_open_wire(dst_ctx *ctx, int measure, int out_kind, size_t max_records, int depth, int wire, dst_stream **out)
{
    if (!ctx || !out)
        return DST_ERR_ARG;
    *out = nullptr;
    if (wire != DST_WIRE_CODES && wire != DST_WIRE_NIBBLES)
        return fail(ctx, DST_ERR_ARG, "unknown wire format");
    if (measure < DST_N || measure > DST_TN93)
        return fail(ctx, DST_ERR_ARG, "unknown measure");
    if (out_kind != DST_OUT_DISTANCE && out_kind != DST_OUT_TALLY)
        return fail(ctx, DST_ERR_ARG, "a stream delivers DST_OUT_DISTANCE or DST_OUT_TALLY");
    if (max_records == 0 || depth < 2 || depth > 16)
        return fail(ctx, DST_ERR_ARG, "max_records must be positive and depth in 2..16");
    const DeviceSet &loaded = ctx->set[0];
    if (!loaded.loaded)
        return fail(ctx, DST_ERR_STATE, "upload the loaded set to slot 0 before opening a stream");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    dst_stream *s = new (std::nothrow) dst_stream;
    if (!s)
        return DST_ERR_NOMEM;
    s->ctx = ctx;
    s->measure = measure;
    s->out_kind = out_kind;
    s->max_records = max_records;
    s->len = loaded.len;
    s->wire = wire;
    // rows 128 bytes apart at least: whole 128-site chunks of input per row (64 bytes of nibbles, 128 of codes)
    s->pitch = wire == DST_WIRE_NIBBLES ? std::max<size_t>(((((loaded.len + 127) / 128) * 64 + 127) / 128) * 128, 128)
                                        : std::max<size_t>(((loaded.len + 127) / 128) * 128, 128);
    s->out_bytes_per_record = dst_out_bytes(measure, out_kind, loaded.n);
    s->slots.resize((size_t)depth);
    hipError_t e = hipStreamCreateWithFlags(&s->s_in, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->s_compute, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->s_out, hipStreamNonBlocking);
    const size_t in_bytes = counts_offset(s) + max_records * 16;
    const size_t out_bytes = std::max<size_t>(s->out_bytes_per_record * max_records, 16);
    for (auto &sl : s->slots) {
        if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_in, in_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_in, in_bytes);
        if (e == hipSuccess) e = hipMalloc(&sl.d_out, out_bytes);
        if (e == hipSuccess) e = hipHostMalloc(&sl.h_out, out_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_bad, sizeof(unsigned long long));
        if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_bad, sizeof(unsigned long long), hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.h2d, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.computed, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.landed, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        destroy(s);
        return fail_hip(ctx, e, "dst_stream_open");
    }
    *out = s;
    return DST_OK;
}

int dst_stream_acquire(dst_stream *s, uint8_t **codes, size_t *pitch, uint32_t **base_counts)
{
    if (!s || !codes || !pitch)
        return DST_ERR_ARG;
    if (s->acquired >= 0)
        return fail(s->ctx, DST_ERR_STATE, "a buffer is already acquired: submit it first");
    auto &sl = s->slots[s->next_acquire % s->slots.size()];
    if (sl.state == 2)
        return fail(s->ctx, DST_ERR_STATE, "every slot is in flight: collect a batch first");
    // the slot's previous batch: its kernels read the slot's planes, its copies read the slot's buffers
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    if (sl.landed && sl.state == 3)
        HIP_TRY(s->ctx, hipEventSynchronize(sl.landed));
    sl.state = 1;
    s->acquired = (int)(s->next_acquire % s->slots.size());
    *codes = sl.h_in;
    *pitch = s->pitch;
    if (base_counts)
        *base_counts = reinterpret_cast<uint32_t *>(sl.h_in + counts_offset(s));
    return DST_OK;
}

int dst_stream_submit(dst_stream *s, size_t n_records, int use_base_counts)
{
    if (!s)
        return DST_ERR_ARG;
    dst_ctx *ctx = s->ctx;
    if (s->acquired < 0)
        return fail(ctx, DST_ERR_STATE, "no buffer acquired");
    if (n_records == 0 || n_records > s->max_records)
        return fail(ctx, DST_ERR_ARG, "n_records must be in 1..max_records");
    DeviceSet &loaded = ctx->set[0];
    if (!loaded.loaded || loaded.len != s->len)
        return fail(ctx, DST_ERR_STATE, "the loaded set changed while the stream was open");
    auto &sl = s->slots[(size_t)s->acquired];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // copy-in: the batch's bytes (and the caller's base counts)
    if (s->len)
        HIP_TRY(ctx, hipMemcpyAsync(sl.d_in, sl.h_in, n_records * s->pitch, hipMemcpyHostToDevice, s->s_in));
    if (use_base_counts)
        HIP_TRY(ctx, hipMemcpyAsync(sl.d_in + counts_offset(s), sl.h_in + counts_offset(s), n_records * 16,
                                    hipMemcpyHostToDevice, s->s_in));
    HIP_TRY(ctx, hipEventRecord(sl.h2d, s->s_in));
    // compute: pack, then the batch (rows) against the loaded set (columns): streamed-major order
    HIP_TRY(ctx, hipStreamWaitEvent(s->s_compute, sl.h2d, 0));
    int rc = pack_queue(ctx, sl.set, sl.d_in, n_records, s->len, s->pitch,
                        use_base_counts ? reinterpret_cast<const uint32_t *>(sl.d_in + counts_offset(s)) : nullptr,
                        sl.d_bad, s->s_compute, false, s->wire == DST_WIRE_NIBBLES);
    if (rc)
        return rc;
    sl.set.loaded = true;  // validity is reported by dst_stream_collect
    if (s->measure == DST_TN93 && s->out_kind == DST_OUT_DISTANCE && !sl.set.have_counts) {
        HIP_TRY(ctx, launch_fill_counts(sl.set, s->s_compute));  // same stream as the kernel that reads them
        sl.set.have_counts = true;
    }
    const size_t bytes = s->out_bytes_per_record * n_records;
    rc = run_sets(ctx, s->measure, false, sl.set, loaded, 0, n_records, s->out_kind, sl.d_out, bytes, (void *)s->s_compute);
    if (rc)
        return rc;
    HIP_TRY(ctx, hipEventRecord(sl.computed, s->s_compute));
    // copy-out
    HIP_TRY(ctx, hipStreamWaitEvent(s->s_out, sl.computed, 0));
    HIP_TRY(ctx, hipMemcpyAsync(sl.h_bad, sl.d_bad, sizeof(unsigned long long), hipMemcpyDeviceToHost, s->s_out));
    if (bytes)
        HIP_TRY(ctx, hipMemcpyAsync(sl.h_out, sl.d_out, bytes, hipMemcpyDeviceToHost, s->s_out));
    HIP_TRY(ctx, hipEventRecord(sl.landed, s->s_out));
    sl.n = n_records;
    sl.state = 2;
    s->acquired = -1;
    s->next_acquire += 1;
    s->in_flight += 1;
    return DST_OK;
}

int dst_stream_collect(dst_stream *s, size_t *n_records, const void **results)
{
    if (!s || !n_records || !results)
        return DST_ERR_ARG;
    dst_ctx *ctx = s->ctx;
    if (s->in_flight == 0)
        return fail(ctx, DST_ERR_STATE, "no submitted batch to collect");
    auto &sl = s->slots[s->next_collect % s->slots.size()];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(sl.landed));
    s->next_collect += 1;
    s->in_flight -= 1;
    sl.state = 3;
    if (*sl.h_bad != ~0ull)
        return invalid_code_error(ctx, *sl.h_bad, s->len);
    *n_records = sl.n;
    *results = sl.h_out;
    return DST_OK;
}

int dst_stream_in_flight(const dst_stream *s) { return s ? (int)s->in_flight : -1; }

int dst_stream_close(dst_stream *s)
{
    destroy(s);
    return DST_OK;
}

}  // extern "C"
