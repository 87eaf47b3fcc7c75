// dst_api.cpp — the GPU-facing half of the C ABI (include/distance_hip.h): context, upload,
// run.  One context = one GPU = one owner thread.  No CPU fallback: every entry point that needs
// the device fails with DST_ERR_HIP when HIP does.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "dst_ctx.h"

using namespace dst;

namespace dst {

std::string g_create_err;
std::mutex g_create_mu;

int publish_prep(dst_ctx *ctx, hipStream_t stream);
int wait_for_other_runs(dst_ctx *ctx, hipStream_t stream);

int fail(dst_ctx *ctx, int status, const std::string &msg)
{
    if (ctx)
        ctx->err = msg;
    return status;
}

int fail_hip(dst_ctx *ctx, hipError_t e, const char *what)
{
    return fail(ctx, e == hipErrorOutOfMemory ? DST_ERR_NOMEM : DST_ERR_HIP,
                std::string(what) + ": " + hipGetErrorString(e));
}

int timer_begin(dst_ctx *ctx, int which, hipStream_t stream)
{
    dst_ctx::Timer &t = ctx->timer[which];
    HIP_TRY(ctx, hipEventRecord(t.begin[t.seq % dst_ctx::kTimerRing], stream));
    return DST_OK;
}

int timer_end(dst_ctx *ctx, int which, hipStream_t stream)
{
    dst_ctx::Timer &t = ctx->timer[which];
    HIP_TRY(ctx, hipEventRecord(t.end[t.seq % dst_ctx::kTimerRing], stream));
    t.seq += 1;
    (which == 0 ? ctx->timed_pair : ctx->timed_pack) = true;
    return DST_OK;
}

int ensure_bytes(dst_ctx *ctx, void **ptr, size_t *have, size_t want)
{
    if (*have >= want)
        return DST_OK;
    if (*ptr) {
        HIP_TRY(ctx, hipFree(*ptr));
        *ptr = nullptr;
        *have = 0;
    }
    HIP_TRY(ctx, hipMalloc(ptr, want));
    *have = want;
    return DST_OK;
}

void free_set(DeviceSet &s)
{
    void *bufs[] = {s.planes, s.counts, s.ref.planes, s.ref.hot_planes, s.ref.hot_sites, s.ref.stats, s.ref.partials, s.rec.off, s.rec.ent, s.rec.range_start, s.rec.pre_cold, s.rec.pre_slots,
                    s.site.inl, s.site.ent, s.aconst, s.runs.index, s.runs.mask, s.runs.known, s.runs.panel_first, s.runs.state,
                    s.runs.aent, s.runs.corr, s.runs.corr_t, s.runs.s7};   // (runs.cnt_run / run_cold / run_hot / ids live in the pre_cold and index blocks)
    for (void *b : bufs)
        if (b)
            (void)hipFree(b);
    if (s.hot) {
        free_set(*s.hot);
        delete s.hot;
    }
    const uint64_t epoch = s.epoch;
    s = DeviceSet{};
    s.epoch = epoch + 1;  // lists other sets hold against this set's reference are stale from now on
}

// (re)allocate the planes / counts of a set for n x len
int shape_set(dst_ctx *ctx, DeviceSet &s, size_t n, size_t len)
{
    // an empty alignment (len == 0) still gets one all-N chunk so every kernel has something to read
    const size_t nchunks = std::max<size_t>(1, (len + kChunkSites - 1) / kChunkSites);
    const size_t npad = ((n + 256 + kPadRecords - 1) / kPadRecords) * kPadRecords;
    const size_t bytes = (size_t)PL_COUNT * nchunks * npad * sizeof(uint4);
    // (a set keeps the largest planes it ever needed: the batches of a stream differ in size, and hipFree would stall
    // every stream of the device — the layout depends on npad and nchunks, the allocation only on their product)
    if (!s.planes || !s.counts || s.planes_bytes < bytes || s.counts_cap < npad) {
        free_set(s);
        HIP_TRY(ctx, hipMalloc((void **)&s.planes, bytes ? bytes : 16));
        s.planes_bytes = bytes;
        const hipError_t e = hipMalloc((void **)&s.counts, npad * 4 * sizeof(uint32_t));
        if (e != hipSuccess) {
            free_set(s);  // never leave a set with planes but no counts behind
            return fail_hip(ctx, e, "hipMalloc(base counts)");
        }
        s.counts_cap = npad;
    }
    s.n = n;
    s.len = len;
    s.nchunks = nchunks;
    s.npad = npad;
    s.loaded = false;
    s.have_counts = false;
    s.lean = false;
    s.planes_deferred = false;
    s.partial = false;
    s.part_begin = s.part_end = 0;
    s.epoch += 1;  // new contents: the reference and the difference lists are rebuilt on demand
    s.ref.valid = s.rec.valid = s.site.valid = s.rec.pre_valid = s.rec.ranges_valid = false;
    s.aconst_family = -1;
    s.runs.active = false;
    s.runs.n_run = 0;
    s.runs.corr_family = -1;
    return DST_OK;
}

// buffers of a set's reference sequence (dst_consensus.hip)
int alloc_ref(dst_ctx *ctx, DeviceSet &s)
{
    if (s.ref.nchunks == s.nchunks && s.ref.planes)
        return DST_OK;
    for (void *b : {(void *)s.ref.planes, (void *)s.ref.hot_planes, (void *)s.ref.hot_sites, (void *)s.ref.stats, (void *)s.ref.partials})
        if (b)
            HIP_TRY(ctx, hipFree(b));
    s.ref.planes = nullptr;
    s.ref.hot_planes = nullptr;
    s.ref.hot_sites = nullptr;
    s.ref.stats = nullptr;
    s.ref.partials = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&s.ref.planes, 4 * s.nchunks * sizeof(uint4)));
    HIP_TRY(ctx, hipMalloc((void **)&s.ref.hot_planes, s.nchunks * sizeof(uint4)));
    HIP_TRY(ctx, hipMalloc((void **)&s.ref.hot_sites, s.nchunks * kChunkSites * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc((void **)&s.ref.stats, 8 * sizeof(uint64_t)));
    HIP_TRY(ctx, hipMalloc((void **)&s.ref.partials, s.nchunks * 16 * sizeof(uint32_t)));
    s.ref.nchunks = s.nchunks;
    return DST_OK;
}

bool planes_deferred_by_pack()
{
    static const bool keep = std::getenv("DST_PACK_ALL_PLANES") != nullptr;   // measurement knob: the pack as it was before r03
    return !keep;
}

// queue the pack of an n x len byte matrix (device memory) into `s`; *d_first_bad receives the first
// offending byte's index (or stays ~0).  Nothing here waits for the device.
// want_lists: the consensus path is likely to run on this set — its reference sequence is sampled from the bytes
// first and the pack counts every record's differences from it on the way (pre_cold / pre_hot), which is the
// consensus path's first pass over the planes; the caller's one synchronisation then also brings the statistics.
int pack_queue(dst_ctx *ctx, DeviceSet &s, const uint8_t *d_codes, size_t n, size_t len, size_t row_stride,
               const uint32_t *d_counts, unsigned long long *d_first_bad, hipStream_t stream, bool want_lists, bool nibbles)
{
    if (nibbles)
        want_lists = false;   // (the 4-bit wire format serves streamed batches: small row sets, lists built on demand)
    int rc = shape_set(ctx, s, n, len);
    if (rc)
        return rc;
    PackLists pl{};
    if (want_lists) {
        rc = alloc_ref(ctx, s);
        if (!rc && s.rec.pre_cap < n + 1) {
            // one block: [n + 1] cold counts, [n + 1] hot counts, the two totals, then the run-chunk counters (RunIndex):
            // [n + 1] run chunks, [n + 1] their cold entries, [n + 1] their hot entries — one allocation, cleared per upload
            if (s.rec.pre_cold)
                HIP_TRY(ctx, hipFree(s.rec.pre_cold));
            s.rec.pre_cold = s.rec.pre_hot = nullptr;
            s.rec.pre_totals = nullptr;
            s.rec.pre_cap = 0;
            const size_t words = 5 * (n + 1) + 8 + ((2 * (n + 1)) & 1);   // four 64-bit totals on an 8-byte boundary
            HIP_TRY(ctx, hipMalloc((void **)&s.rec.pre_cold, words * sizeof(uint32_t)));
            s.rec.pre_cap = n + 1;
        }
        if (!rc && s.runs.n_alloc < n + 1) {   // the run records' numbers and ids, and what the report kernel decides
            if (s.runs.index)
                HIP_TRY(ctx, hipFree(s.runs.index));
            s.runs.index = s.runs.ids = nullptr;
            s.runs.n_alloc = 0;
            HIP_TRY(ctx, hipMalloc((void **)&s.runs.index, 2 * (n + 1) * sizeof(uint32_t)));
            s.runs.n_alloc = n + 1;
            if (!s.runs.state)
                HIP_TRY(ctx, hipMalloc((void **)&s.runs.state, 4 * sizeof(uint32_t)));
        }
        if (!rc)
            rc = ensure_bytes(ctx, (void **)&s.rec.pre_slots, &s.rec.pre_slots_cap, s.nchunks * s.npad * sizeof(uint4));
        if (rc)
            return rc;
        {
            const size_t cap = s.rec.pre_cap, pad = (2 * cap) & 1;
            s.rec.pre_hot = s.rec.pre_cold + cap;
            s.rec.pre_totals = reinterpret_cast<unsigned long long *>(s.rec.pre_cold + 2 * cap + pad);
            s.runs.cnt_run = s.rec.pre_cold + 2 * cap + pad + 8;
            s.runs.run_cold = s.runs.cnt_run + cap;
            s.runs.run_hot = s.runs.run_cold + cap;
            s.runs.ids = s.runs.index + s.runs.n_alloc;
            // (the pack's counts and first-invalid-byte cell are cleared on the sample's way: no fills of their own)
            HIP_TRY(ctx, launch_ref_sample_bytes(d_codes, row_stride, s, stream, s.rec.pre_cold, 5 * cap + pad + 8, d_first_bad));
            HIP_TRY(ctx, launch_hot_list(s, stream));
        }
        pl.ref_planes = s.ref.planes;
        pl.hot_planes = s.ref.hot_planes;
        pl.stats = reinterpret_cast<const unsigned long long *>(s.ref.stats);
        pl.max_dev_sum = (unsigned long long)(kListsMaxDeviation * (double)len * (double)std::min<size_t>(n, kRefSamples));
        pl.cnt_cold = s.rec.pre_cold;
        pl.cnt_hot = s.rec.pre_hot;
        pl.slots = s.rec.pre_slots;
        pl.defer_planes = planes_deferred_by_pack() ? 1 : 0;
        static const bool no_runs = std::getenv("DST_NO_RUN_RECORDS") != nullptr;   // measurement knob: r02's lists
        if (!no_runs) {
            pl.cnt_run = s.runs.cnt_run;
            pl.run_cold = s.runs.run_cold;
            pl.run_hot = s.runs.run_hot;
        }
    }
    if (!want_lists)
        HIP_TRY(ctx, hipMemsetAsync(d_first_bad, 0xFF, sizeof(unsigned long long), stream));
    if (int rc_t = timer_begin(ctx, 1, stream))
        return rc_t;
    if (nibbles)
        HIP_TRY(ctx, launch_pack_nibbles(d_codes, row_stride, s, d_first_bad, stream));
    else
        HIP_TRY(ctx, launch_pack(d_codes, row_stride, s, d_first_bad, want_lists ? &pl : nullptr, stream));
    if (int rc_t = timer_end(ctx, 1, stream))
        return rc_t;
    if (d_counts) {
        HIP_TRY(ctx, hipMemsetAsync(s.counts, 0, s.npad * 4 * sizeof(uint32_t), stream));
        HIP_TRY(ctx, hipMemcpyAsync(s.counts, d_counts, n * 4 * sizeof(uint32_t),
                                    hipMemcpyDeviceToDevice, stream));
        s.have_counts = true;
    }
    return DST_OK;
}

int invalid_code_error(dst_ctx *ctx, unsigned long long first_bad, size_t len)
{
    char msg[160];
    std::snprintf(msg, sizeof msg,
                  "invalid nucleotide code in record %llu at site %llu (not a value src/encoding.rs produces)",
                  first_bad / (len ? len : 1), first_bad % (len ? len : 1));
    return fail(ctx, DST_ERR_INVALID_CODE, msg);
}

bool consensus_shape_ok(const DeviceSet &rows, const DeviceSet &cols);

int pack_from_device(dst_ctx *ctx, int slot, const uint8_t *d_codes, size_t n, size_t len,
                     size_t row_stride, const uint32_t *d_counts, hipStream_t stream)
{
    DeviceSet &s = ctx->set[slot];
    // the consensus path's preparation rides on the pack when a run of this set is likely to take it: not forced
    // dense, a shape the lists can index, and more work than the dense kernels finish before lists are built
    const bool want_lists = ctx->path != DST_PATH_DENSE && n >= 2 && len > 0 && n < kEntryMask && len < kSiteMask &&
                            0.5 * (double)n * (double)n * (double)len >= ctx->prep_min_work;
    int rc = pack_queue(ctx, s, d_codes, n, len, row_stride, d_counts, ctx->d_first_bad, stream, want_lists);
    if (rc)
        return rc;
    // the device writes its report (first invalid byte, statistics, totals) into page-locked host memory: one wait
    // run records (RunIndex): at most a third of the records, and correction tables (two, up to four words each) below
    // 24 GB of the 288 (6 GB until r03: 50,000 records of which a fifth carry runs of N lost their run records to it and
    // took 35 ms where 20,000 such records took 1.7)
    const uint32_t max_run = (uint32_t)std::min<uint64_t>(n / 3, 24000000000ull / (32ull * std::max<size_t>(n, 1)));
    HIP_TRY(ctx, launch_report(ctx->d_first_bad,
                               want_lists ? reinterpret_cast<const unsigned long long *>(s.ref.stats) : nullptr,
                               want_lists ? s.rec.pre_cold : nullptr, want_lists ? s.rec.pre_hot : nullptr, n, ctx->d_report, stream,
                               want_lists ? &s.runs : nullptr, max_run, want_lists ? s.rec.pre_totals : nullptr));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    const unsigned long long first_bad = ctx->h_report[0], totals[2] = {ctx->h_report[9], ctx->h_report[10]};
    if (want_lists)
        for (int k = 0; k < 8; ++k)
            s.ref.h_stats[k] = ctx->h_report[1 + k];
    if (first_bad != ~0ull)
        return invalid_code_error(ctx, first_bad, len);
    s.loaded = true;
    if (want_lists) {
        s.ref.valid = true;
        const double max_dev = kListsMaxDeviation * (double)len * (double)std::min<size_t>(n, kRefSamples);
        s.rec.pre_valid = (double)s.ref.h_stats[1] <= (double)(unsigned long long)max_dev;   // the kernel's own test
        s.lean = s.rec.pre_valid;   // ... which also made it skip the four derived planes
        s.planes_deferred = s.rec.pre_valid && planes_deferred_by_pack();   // ... and the base planes of every inline chunk
        s.rec.pre_epoch = s.epoch;
        s.rec.pre_total_cold = totals[0];
        s.rec.pre_total_hot = totals[1];
        s.runs.n_run = s.rec.pre_valid ? (uint32_t)ctx->h_report[11] : 0u;
        s.runs.removed = s.runs.n_run ? ctx->h_report[12] : 0;
        s.runs.active = s.runs.n_run != 0;
        s.runs.corr_family = -1;
    }
    return DST_OK;
}

// Whatever reads a set's planes (the dense pair kernels, the hot columns' compaction, lists against another set's
// reference, the consensus and differences calls) first has the base planes the pack deferred written, from the slots and
// the reference they were taken against (planes_from_slots_kernel).  The consensus path never comes here.
int ensure_planes(dst_ctx *ctx, DeviceSet &s, hipStream_t stream)
{
    if (!s.planes_deferred)
        return DST_OK;
    if (s.partial)
        return fail(ctx, DST_ERR_STATE, "a set uploaded with dst_upload_shared runs on the consensus path only (its planes are "
                                        "not stored)");
    int rc = wait_for_other_runs(ctx, stream);
    if (rc)
        return rc;
    HIP_TRY(ctx, launch_planes_from_slots(s, stream));
    s.planes_deferred = false;
    return publish_prep(ctx, stream);
}

// the dense pair kernels read all eight planes: build the four derived ones of a set that was packed lean
int ensure_derived(dst_ctx *ctx, DeviceSet &s, hipStream_t stream)
{
    if (s.partial)
        return fail(ctx, DST_ERR_STATE, "a set uploaded with dst_upload_shared runs on the consensus path only (this rank holds "
                                        "the planes of its own records)");
    int rc = ensure_planes(ctx, s, stream);
    if (rc || !s.lean)
        return rc;
    rc = wait_for_other_runs(ctx, stream);
    if (rc)
        return rc;
    HIP_TRY(ctx, launch_derive(s, stream));
    s.lean = false;
    return publish_prep(ctx, stream);
}

int need_counts(dst_ctx *ctx, DeviceSet &s, hipStream_t stream)
{
    if (s.have_counts)
        return DST_OK;
    if (s.partial)
        return fail(ctx, DST_ERR_STATE, "the set was uploaded with dst_upload_shared without base counts (with_counts = 0): this rank "
                                        "holds the planes of its own records only");
    if (s.planes_deferred) {   // a chunk that is inline in its slot is counted from the slot
        PackLists pl{};
        pl.slots = s.rec.pre_slots;
        pl.ref_planes = s.ref.planes;
        pl.defer_planes = 1;
        HIP_TRY(ctx, launch_fill_counts(s, stream, &pl));
    } else {
        HIP_TRY(ctx, launch_fill_counts(s, stream));
    }
    // later runs may be queued on OTHER streams (multi-GPU sub-slabs alternate between two): they wait for this on
    // the device (order_after_prep)
    s.have_counts = true;
    return publish_prep(ctx, stream);
}

// Tile lists already on the device, keyed by the launch geometry (no host sync or H2D on a hit).
// bn >= 0: dense BlockDesc list of tile shape (bm, bn); bn == -1: consensus-path tiles of bm rows.
int prepare_schedule(dst_ctx *ctx, bool square, uint64_t rb, uint64_t re, uint64_t ncols, int bm, int bn,
                     hipStream_t stream, const void **d_blocks, uint32_t *nblocks)
{
    constexpr size_t kMaxSchedules = 16;
    for (auto &s : ctx->schedules) {
        if (s.square == square && s.rb == rb && s.re == re && s.ncols == ncols && s.bm == bm && s.bn == bn) {
            s.last_use = ++ctx->schedule_clock;
            if (std::find(s.users.begin(), s.users.end(), stream) == s.users.end())
                s.users.push_back(stream);
            *d_blocks = s.d_blocks;
            *nblocks = s.nblocks;
            return DST_OK;
        }
    }
    std::vector<BlockDesc> blocks;
    std::vector<ConsensusTile> tiles;
    const void *src = nullptr;
    size_t bytes = 0, count = 0;
    if (bn >= 0) {
        blocks = build_blocks(square, rb, re, ncols, TileShape{bm, bn});
        src = blocks.data();
        count = blocks.size();
        bytes = count * sizeof(BlockDesc);
    } else {
        tiles = build_consensus_tiles(square, rb, re, ncols, (uint32_t)bm);
        src = tiles.data();
        count = tiles.size();
        bytes = count * sizeof(ConsensusTile);
    }
    dst_ctx::Schedule s;
    s.square = square;
    s.rb = rb;
    s.re = re;
    s.ncols = ncols;
    s.bm = bm;
    s.bn = bn;
    s.nblocks = (uint32_t)count;
    s.last_use = ++ctx->schedule_clock;
    if (ctx->schedules.size() >= kMaxSchedules) {
        // evict the least recently used; a kernel queued on any of the streams that launched with it may still read it
        // (sub-slab launches alternate between streams, the caller may bring its own): those streams are waited for — not
        // the device: a loop over row slabs (the CLI's text calls) evicts on every call.  A stream that is gone: the device.
        size_t victim = 0;
        for (size_t k = 1; k < ctx->schedules.size(); ++k)
            if (ctx->schedules[k].last_use < ctx->schedules[victim].last_use)
                victim = k;
        for (hipStream_t u : ctx->schedules[victim].users)
            if (hipStreamSynchronize(u) != hipSuccess) {
                (void)hipGetLastError();
                HIP_TRY(ctx, hipDeviceSynchronize());
                break;
            }
        // its buffer is recycled when it is big enough (row slabs of one run have similar tile counts)
        void *spare = ctx->schedules[victim].d_blocks;
        const size_t spare_bytes = ctx->schedules[victim].bytes;
        ctx->schedules.erase(ctx->schedules.begin() + (long)victim);
        if (spare && spare_bytes >= bytes && count) {
            s.d_blocks = spare;
            s.bytes = spare_bytes;
        } else if (spare) {
            HIP_TRY(ctx, hipFree(spare));
        }
    }
    if (count) {
        if (!s.d_blocks) {
            s.bytes = bytes + bytes / 4;
            HIP_TRY(ctx, hipMalloc(&s.d_blocks, s.bytes));
        }
        // pageable source: the copy is complete on return, later kernels on any stream see it
        HIP_TRY(ctx, hipMemcpy(s.d_blocks, src, bytes, hipMemcpyHostToDevice));
    }
    s.users.assign(1, stream);
    ctx->schedules.push_back(s);
    *d_blocks = s.d_blocks;
    *nblocks = s.nblocks;
    return DST_OK;
}

int prepare_blocks(dst_ctx *ctx, bool square, uint64_t rb, uint64_t re, uint64_t ncols, TileShape ts,
                   hipStream_t stream, const BlockDesc **d_blocks, uint32_t *nblocks)
{
    const void *p = nullptr;
    const int rc = prepare_schedule(ctx, square, rb, re, ncols, ts.bm, ts.bn, stream, &p, nblocks);
    *d_blocks = static_cast<const BlockDesc *>(p);
    return rc;
}

// =============================================================================================
// consensus-delta path: reference, difference lists, path choice (kernels: dst_consensus.hip)
// =============================================================================================
constexpr uint64_t kMaxListEntries = 0x7FFFFFFFull;  // 32-bit CSR offsets
constexpr uint32_t kConsensusRowsPerTile = kTileRowsMax;

bool consensus_shape_ok(const DeviceSet &rows, const DeviceSet &cols)
{
    // list entries carry a site or a record in 28 bits next to the nibble
    return rows.n < kEntryMask && cols.n < kEntryMask && rows.len < kSiteMask && rows.len == cols.len && rows.len > 0;
}

int ensure_lut(dst_ctx *ctx)
{
    if (ctx->d_lut)
        return DST_OK;
    auto lut = std::make_unique<ConsensusLut>();
    build_consensus_lut(*lut);
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_lut, sizeof(ConsensusLut)));
    HIP_TRY(ctx, hipMemcpy(ctx->d_lut, lut.get(), sizeof(ConsensusLut), hipMemcpyHostToDevice));
    if (!ctx->d_total)   // (the text path may have made it already)
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_total, 2 * sizeof(unsigned long long)));  // [0] list entries, [1] overflow entries
    return DST_OK;
}

// ---- ordering between streams without host synchronisation -----------------------------------------
// Derived data (difference lists, per-record constants, hot columns, base counts) is built on the stream of the run
// that first needs it.  A later run on ANOTHER stream waits for prep_event on the device; a rebuild first waits,
// on the device, for the runs other streams still have in flight.  (r02 first used hipDeviceSynchronize /
// hipStreamSynchronize here: six host round trips per step, 0.4 ms of a 0.9 ms step at 10,000 x 30,000.)
int publish_prep(dst_ctx *ctx, hipStream_t stream)
{
    HIP_TRY(ctx, hipEventRecord(ctx->prep_event, stream));
    ctx->prep_stream = stream;
    ctx->prep_pending = true;
    static const bool host_sync = std::getenv("DST_HOST_SYNC") != nullptr;  // measurement knob: the r02a behaviour
    if (host_sync)
        HIP_TRY(ctx, hipStreamSynchronize(stream));
    return DST_OK;
}

int order_after_prep(dst_ctx *ctx, hipStream_t stream)
{
    if (ctx->prep_pending && ctx->prep_stream != stream)
        HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->prep_event, 0));
    return DST_OK;
}

int wait_for_other_runs(dst_ctx *ctx, hipStream_t stream)
{
    for (auto &r : ctx->recent)
        if (r.used && r.stream != stream)
            HIP_TRY(ctx, hipStreamWaitEvent(stream, r.event, 0));
    return DST_OK;
}

int note_run(dst_ctx *ctx, hipStream_t stream)
{
    dst_ctx::Recent *slot = nullptr;
    for (auto &r : ctx->recent)
        if (r.used && r.stream == stream)
            slot = &r;
    if (!slot) {
        slot = &ctx->recent[ctx->recent_next % 4];
        ctx->recent_next += 1;
    }
    HIP_TRY(ctx, hipEventRecord(slot->event, stream));
    slot->stream = stream;
    slot->used = true;
    return DST_OK;
}

// the reference sequence of `s` (plurality code per site over a sample of its records), its hot sites and the
// statistics the path choice reads
int ensure_ref(dst_ctx *ctx, DeviceSet &s, hipStream_t stream)
{
    if (s.ref.valid)
        return DST_OK;
    int rc_alloc = alloc_ref(ctx, s);
    if (rc_alloc)
        return rc_alloc;
    if (int rc_p = ensure_planes(ctx, s, stream))   // (the sample reads planes)
        return rc_p;
    HIP_TRY(ctx, launch_ref_sample(s, stream));
    HIP_TRY(ctx, launch_hot_list(s, stream));
    HIP_TRY(ctx, hipMemcpyAsync(s.ref.h_stats, s.ref.stats, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    s.ref.valid = true;
    return DST_OK;
}

// hybrid path: the hot columns of `s` (hot by refset's reference) as a packed set of their own
int ensure_hot(dst_ctx *ctx, DeviceSet &s, DeviceSet &refset, hipStream_t stream)
{
    const size_t n_hot = refset.ref.h_stats[4];
    if (s.hot && s.hot->loaded && s.hot_epoch == s.epoch && s.hot_ref_owner == &refset && s.hot_ref_epoch == refset.epoch)
        return DST_OK;
    int rc0 = wait_for_other_runs(ctx, stream);  // a run on another stream may still read the old columns
    if (rc0)
        return rc0;
    if (!s.hot)
        s.hot = new (std::nothrow) DeviceSet;
    if (!s.hot)
        return DST_ERR_NOMEM;
    int rc = shape_set(ctx, *s.hot, s.n, n_hot);
    if (rc)
        return rc;
    rc = ensure_planes(ctx, s, stream);   // (the hot columns are gathered from the planes)
    if (rc)
        return rc;
    HIP_TRY(ctx, launch_compact(s, refset.ref.hot_sites, (uint32_t)n_hot, *s.hot, stream));
    rc = publish_prep(ctx, stream);
    if (rc)
        return rc;
    s.hot->loaded = true;
    s.hot_epoch = s.epoch;
    s.hot_ref_owner = &refset;
    s.hot_ref_epoch = refset.epoch;
    return DST_OK;
}

// Difference lists of `s` against the reference of `refset` (and, for a column set, the same entries by
// site and panel).  DST_ERR_CAPACITY: more entries than 32-bit offsets hold — the caller runs dense.
int ensure_index(dst_ctx *ctx, DeviceSet &s, DeviceSet &refset, bool want_sites, bool without_hot, hipStream_t stream)
{
    const bool lists_ok = s.rec.valid && s.rec.ref_owner == &refset && s.rec.ref_epoch == refset.epoch &&
                          s.rec.without_hot == without_hot;
    if (lists_ok && (!want_sites || s.site.valid))
        return DST_OK;
    // other streams may still be reading the buffers about to be rebuilt
    int rc = wait_for_other_runs(ctx, stream);
    if (rc)
        return rc;
    const uint32_t n_panels = (uint32_t)((s.n + kPanelCols - 1) / kPanelCols);
    const size_t n_buckets = want_sites ? s.nchunks * kChunkSites * (size_t)n_panels : 0;
    // 32 bytes of lookup table per (site, panel); bucket numbers travel as 32-bit values
    if (n_buckets >= 0xFFFFFFFFull || n_buckets * 2 * sizeof(uint4) > (16ull << 30))
        return fail(ctx, DST_ERR_CAPACITY, "too many sites x panels for the consensus path's lookup table");
    if (lists_ok && s.rec.ranges_valid) {
        // the lists are there with their range marks (they came by dst_upload_shared's exchange): only the site buckets
        rc = ensure_bytes(ctx, (void **)&s.site.inl, &s.site.inl_cap, std::max<size_t>(n_buckets, 1) * 2 * sizeof(uint4));
        if (!rc)
            rc = ensure_bytes(ctx, (void **)&s.site.ent, &s.site.ent_cap, std::max<size_t>(s.rec.total, 1) * sizeof(uint32_t));
        if (rc)
            return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_total, 0, 2 * sizeof(unsigned long long), stream));
        HIP_TRY(ctx, launch_site_buckets(s, n_panels, reinterpret_cast<uint32_t *>(ctx->d_total + 1), stream));
        rc = publish_prep(ctx, stream);
        if (rc)
            return rc;
        s.site.valid = true;
        s.site.n_panels = n_panels;
        return DST_OK;
    }
    if (s.partial)
        return fail(ctx, DST_ERR_STATE, "a set uploaded with dst_upload_shared holds the lists the exchange brought; they cannot be "
                                        "rebuilt here (another reference, the hybrid path): upload it with dst_upload_device");
    rc = ensure_bytes(ctx, (void **)&s.rec.off, &s.rec.off_cap, (s.n + 1) * sizeof(uint32_t));
    if (!rc && want_sites)
        rc = ensure_bytes(ctx, (void **)&s.site.inl, &s.site.inl_cap, std::max<size_t>(n_buckets, 1) * 2 * sizeof(uint4));
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&ctx->scan_tmp, &ctx->scan_tmp_bytes, scan_tmp_words(s.n + 1) * sizeof(uint32_t));
    if (rc)
        return rc;
    s.rec.valid = false;
    s.rec.ranges_valid = false;
    s.site.valid = false;
    uint32_t *d_ovf_n = reinterpret_cast<uint32_t *>(ctx->d_total + 1);
    const uint4 *hot_planes = without_hot ? refset.ref.hot_planes : nullptr;
    unsigned long long total = 0;
    const uint32_t *scan_src0 = nullptr, *scan_src1 = nullptr;
    const bool from_pack = &s == &refset && s.rec.pre_valid && s.rec.pre_epoch == s.epoch;
    if (!from_pack) {
        s.runs.active = false;   // lists from the planes keep every entry (index_kernel knows no run chunks)
        rc = ensure_planes(ctx, s, stream);
        if (rc)
            return rc;
    }
    if (from_pack) {
        // the pack counted the list lengths against this very reference: no pass over the planes, no round trip
        // (the scan below reads the counts where the pack left them)
        total = s.rec.pre_total_cold;
        if (!without_hot && s.rec.pre_total_hot) {
            scan_src1 = s.rec.pre_hot;
            total += s.rec.pre_total_hot;
        }
        scan_src0 = s.rec.pre_cold;
    } else {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_total, 0, 2 * sizeof(unsigned long long), stream));
        HIP_TRY(ctx, hipMemsetAsync(s.rec.off, 0, (s.n + 1) * sizeof(uint32_t), stream));
        HIP_TRY(ctx, launch_index(s, refset.ref.planes, hot_planes, false, false, s.rec.off, nullptr, ctx->d_total, stream));
        HIP_TRY(ctx, hipMemcpyAsync(&total, ctx->d_total, sizeof total, hipMemcpyDeviceToHost, stream));
        HIP_TRY(ctx, hipStreamSynchronize(stream));
    }
    if (total > kMaxListEntries / 4)
        return fail(ctx, DST_ERR_CAPACITY, "too many differences from the reference sequence for the consensus path");
    s.rec.total = total;
    const size_t cap = std::max<size_t>(total, 1);
    rc = ensure_bytes(ctx, (void **)&s.rec.ent, &s.rec.ent_cap, (cap + 4) * sizeof(uint32_t));  // 16-byte reads past the end
    if (!rc && want_sites)
        rc = ensure_bytes(ctx, (void **)&s.site.ent, &s.site.ent_cap, cap * sizeof(uint32_t));
    if (!rc && want_sites)
        rc = ensure_bytes(ctx, (void **)&s.rec.range_start, &s.rec.range_cap,
                          ((s.nchunks * kChunkSites + kBucketSites - 1) / kBucketSites) * s.npad * sizeof(uint32_t));
    if (rc)
        return rc;
    // (the scan also clears the counters behind it: [0..1] the list total of the count pass, [2..3] the overflow entries)
    HIP_TRY(ctx, launch_exclusive_scan(s.rec.off, s.n + 1, ctx->scan_tmp, stream, scan_src0, scan_src1,
                                       reinterpret_cast<uint32_t *>(ctx->d_total), 4));
    if (from_pack)   // the entries are in the pack's slots already
        HIP_TRY(ctx, launch_slot_fill(s, refset.ref.planes, refset.ref.hot_planes, without_hot, s.rec.off, s.rec.ent,
                                      want_sites ? s.rec.range_start : nullptr, stream));
    else
        HIP_TRY(ctx, launch_index(s, refset.ref.planes, hot_planes, true, false, s.rec.off, s.rec.ent, ctx->d_total, stream,
                                  want_sites ? s.rec.range_start : nullptr));
    if (want_sites)
        HIP_TRY(ctx, launch_site_buckets(s, n_panels, d_ovf_n, stream));
    if (from_pack && s.runs.active) {
        // run records: their run chunks as bit masks (from the slots' flags) and the known reference sites per chunk, for
        // these lists' flavour (the hybrid path's lists leave the hot sites to the dense kernels)
        RunIndex &ru = s.runs;
        ru.mask_words = (s.nchunks + 31) / 32;
        rc = ensure_bytes(ctx, (void **)&ru.mask, &ru.mask_cap, 2 * (size_t)ru.n_run * ru.mask_words * sizeof(uint32_t));   // by record, then transposed
        if (!rc)
            rc = ensure_bytes(ctx, (void **)&ru.known, &ru.known_cap, ((s.nchunks + 31) / 32 * 32) * sizeof(uint32_t));
        if (!rc)
            rc = ensure_bytes(ctx, (void **)&ru.panel_first, &ru.panel_cap, ((size_t)n_panels + 2) * sizeof(uint32_t));
        if (rc)
            return rc;
        HIP_TRY(ctx, launch_run_masks(s, stream));
        HIP_TRY(ctx, launch_run_known(s, refset.ref.planes, hot_planes, stream));
        ru.corr_family = -1;
    }
    // runs queued on other streams wait for this on the device
    rc = publish_prep(ctx, stream);
    if (rc)
        return rc;
    s.rec.valid = true;
    s.rec.ranges_valid = want_sites;
    s.rec.ref_owner = &refset;
    s.rec.ref_epoch = refset.epoch;
    s.rec.without_hot = without_hot;
    s.aconst_family = -1;
    if (want_sites) {
        s.site.valid = true;
        s.site.n_panels = n_panels;
    }
    return DST_OK;
}

// A_k words of every record of `s` for (family, packing), summed from its current lists
int ensure_aconst(dst_ctx *ctx, DeviceSet &s, DeviceSet &refset, int family, bool wide, hipStream_t stream)
{
    if (s.aconst && s.aconst_family == family && s.aconst_wide == wide && s.aconst_epoch == s.epoch &&
        s.aconst_ref_owner == &refset && s.aconst_ref_epoch == refset.epoch)
        return DST_OK;
    int rc = wait_for_other_runs(ctx, stream);  // a run on another stream may still read the old words
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&s.aconst, &s.aconst_cap, s.npad * kMaxWords * sizeof(uint32_t));
    if (rc)
        return rc;
    if (s.runs.active) {
        // the entries' a-words (what corr_kernel sums) and the two correction tables of this (family, packing)
        RunIndex &ru = s.runs;
        const size_t words = (size_t)family_words(family, wide);
        rc = ensure_bytes(ctx, (void **)&ru.aent, &ru.aent_cap, (size_t)kMaxWords * std::max<size_t>(s.rec.total, 1) * sizeof(uint32_t));
        if (!rc)
            rc = ensure_bytes(ctx, (void **)&ru.corr, &ru.corr_cap, words * ru.n_run * s.n * sizeof(uint32_t));
        if (!rc)
            rc = ensure_bytes(ctx, (void **)&ru.corr_t, &ru.corr_t_cap, words * ru.n_run * s.n * sizeof(uint32_t));
        if (!rc)   // the records' per-chunk sums in 7-bit pieces: the B operand of the correction tables' matrix product
            rc = ensure_bytes(ctx, (void **)&ru.s7, &ru.s7_cap, words * 5 * ((s.n + 31) / 32) * 1024 * ru.mask_words + 64);   // whole tiles of 32 records
        if (rc)
            return rc;
    }
    HIP_TRY(ctx, launch_aconst(s, family, wide, ctx->d_lut, stream));
    if (s.runs.active)
        HIP_TRY(ctx, launch_run_tables(s, family, wide, s.rec.without_hot, ctx->d_lut, stream));
    rc = publish_prep(ctx, stream);
    if (rc)
        return rc;
    s.aconst_family = family;
    s.aconst_wide = wide;
    s.aconst_epoch = s.epoch;
    s.aconst_ref_owner = &refset;
    s.aconst_ref_epoch = refset.epoch;
    return DST_OK;
}

// Which path is cheapest for this launch?  The sampled reference gives, per site, the fraction p_s of records that
// deviate from it: a pair costs the dense path L sites whatever the data; the consensus path one output plus about
// sum_s p_s^2 intersection events; the hybrid path the dense cost of the H hot sites (p_s > 5 %: kHotPermille), one round trip of
// their tallies through HBM, and the events of the cold sites only.  Constants: measured on MI355X, seconds
// (tools/calibrate.py, profiles/r02/consensus_calibration.txt).
// share: this launch's part of the pair space of the two sets.  A run over a row range is one slab of a job that goes
// on to cover the other rows (the CLI's slabs, a rank's sub-slabs): the lists are built once and serve all of them, so
// a slab is charged its share of the build — every slab then decides like the whole job would (charged in full, each
// 4 Mi-pair slab of a 50,000-record run chose the dense kernels: 0.9 ms instead of 0.03).
// Run records (RunIndex) leave their run chunks out of the lists: the sample's statistics, taken before that, over-state
// what the pair kernel will meet.  List lengths shrink by f = kept / (kept + removed) entries, events (both records
// deviate at a site) by about f^2.
double run_scale(const DeviceSet &cols)
{
    if (!cols.runs.active)
        return 1.0;
    const double kept = (double)(cols.rec.pre_total_cold + cols.rec.pre_total_hot);
    return kept / std::max(kept + (double)cols.runs.removed, 1.0);
}

int cheapest_path(const DeviceSet &rows, const DeviceSet &cols, int measure, uint64_t pairs, uint32_t ntiles, double share)
{
    //                                            n       n_high  raw      jc69     k80      tn93
    static const double dense_site_pairs_per_s[6] = {2.9e14, 2.9e14, 1.95e14, 1.92e14, 1.82e14, 1.48e14};
    static const double out_s_per_pair[6] = {2.2e-12, 2.2e-12, 2.3e-12, 3.9e-12, 5.0e-12, 1.55e-11};
    constexpr double event_s = 1.85e-12;  // per event, launches of more than one per pair (1.8-2.3 ps measured; 4.5 before the
                                          // all-waves launch variant)
    const uint64_t *st = cols.ref.h_stats;
    const double S = (double)std::max<uint64_t>(st[3], 1);
    const double n_hot = (double)st[4];
    // the sample's sum of squared deviant counts over-states sum p^2 by about (mean list length) / S
    const double f = run_scale(cols);
    const double mean_list = f * (double)st[1] / S, mean_list_cold = f * (double)st[6] / S;  // differences per record
    const double events = f * f * (double)st[2] / (S * S), events_cold = f * f * (double)st[7] / (S * S);
    const double dense = (double)pairs * (double)cols.len / dense_site_pairs_per_s[measure];
    // building the lists reads four bit-planes twice; walking a row's list costs one bucket lookup per panel
    const bool have_lists = rows.rec.valid && cols.site.valid;
    // (lengths counted and entries slotted by the pack already: one pass over the slots; else two over four planes)
    const bool from_pack = &rows == &cols && cols.rec.pre_valid && cols.rec.pre_epoch == cols.epoch;
    const double build = ((double)(rows.n + cols.n) * (double)cols.len * (from_pack ? 0.1e-12 : 0.5e-12) + 1.5e-4) *
                         std::min(1.0, std::max(share, 1e-3));
    const double walk = (double)ntiles * kConsensusRowsPerTile * 1e-9 / 256.0;
    // run records: their correction tables (one test per list entry and run record, ~5e-14 s) and what the pair kernel
    // adds from them (two words per pair with a run record)
    const double run_tables = cols.runs.active ? ((double)cols.n * cols.runs.n_run * (mean_list * 5e-14 + 2e-12)) * std::min(1.0, std::max(share, 1e-3))
                                                     + (double)pairs * 2.0 * cols.runs.n_run / std::max<double>((double)cols.n, 1.0) * 1.0e-12
                                               : 0.0;
    const double cons = (double)pairs * (out_s_per_pair[measure] + events * event_s) +
                        ((have_lists && !rows.rec.without_hot) ? 0.0 : build) + walk * mean_list + 3e-5 + run_tables;
    double best = std::min(dense, cons);
    int path = cons < dense ? DST_PATH_CONSENSUS : DST_PATH_DENSE;
    if (n_hot > 0 && n_hot * 4 < (double)cols.len) {
        // hot columns: dense kernels over ceil(H/128) chunks (at least the epilogue: one tally per pair written, then read
        // by the consensus kernel: ~8 bytes per pair at ~4 TB/s), plus gathering the columns (8 plane bits per record and site)
        const double hot_dense = (double)pairs * (std::ceil(n_hot / 128.0) * 128.0 / dense_site_pairs_per_s[measure] + 2.2e-12);
        const double gather = (double)(rows.n + cols.n) * n_hot * 2.5e-11 + 1e-4;
        const double hybrid = (double)pairs * (out_s_per_pair[measure] + events_cold * event_s) + hot_dense + gather +
                              ((have_lists && rows.rec.without_hot) ? 0.0 : build) + walk * mean_list_cold + 6e-5 + run_tables;
        if (hybrid < best) {
            best = hybrid;
            path = DST_PATH_HYBRID;
        }
    }
    return path;
}

int run_common(dst_ctx *ctx, int measure, bool square, int row_slot, int col_slot, uint64_t rb,
               uint64_t re, int out_kind, void *d_out, size_t cap, void *stream_v)
{
    if (!ctx)
        return DST_ERR_ARG;
    if (row_slot < 0 || row_slot > 1 || col_slot < 0 || col_slot > 1)
        return fail(ctx, DST_ERR_ARG, "slot must be 0 or 1");
    return run_sets(ctx, measure, square, ctx->set[row_slot], ctx->set[col_slot], rb, re, out_kind, d_out, cap, stream_v);
}

// rows [rb, re) of `rows` against every (square: later) record of `cols` — any two packed sets of this context
int run_sets(dst_ctx *ctx, int measure, bool square, DeviceSet &rows, DeviceSet &cols, uint64_t rb, uint64_t re,
             int out_kind, void *d_out, size_t cap, void *stream_v)
{
    if (measure < DST_N || measure > DST_TN93)
        return fail(ctx, DST_ERR_ARG, "unknown measure");
    if (out_kind != DST_OUT_DISTANCE && out_kind != DST_OUT_TALLY && out_kind != DST_OUT_TALLY16)
        return fail(ctx, DST_ERR_ARG, "unknown output kind");
    if (!rows.loaded || !cols.loaded)
        return fail(ctx, DST_ERR_STATE, "set not uploaded");
    if (rows.len != cols.len) {
        char msg[128];  // src/fastaio.rs:93-95
        std::snprintf(msg, sizeof msg, "Different length sequences in alignment(s): %zu vs %zu", rows.len,
                      cols.len);
        return fail(ctx, DST_ERR_STATE, msg);
    }
    if (rb > re || re > rows.n)
        return fail(ctx, DST_ERR_ARG, "row range out of bounds");
    if (out_kind == DST_OUT_TALLY16 && rows.len > 65535)
        return fail(ctx, DST_ERR_ARG, "DST_OUT_TALLY16 needs alignments shorter than 65,536 sites");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_v ? (hipStream_t)stream_v : ctx->stream;
    {   // data another run's stream derived (lists, constants, counts): wait for it on the device
        const int rc_prep = order_after_prep(ctx, stream);
        if (rc_prep)
            return rc_prep;
    }
    const uint64_t total_pairs = pairs_in_rows(square, cols.n, rb, re);
    const size_t need = dst_out_bytes(measure, out_kind, total_pairs);
    if (need > cap)
        return fail(ctx, DST_ERR_CAPACITY, "output buffer too small for the requested rows");
    ctx->timed_pair = false;
    ctx->pair_ms = 0;
    if (total_pairs == 0)
        return DST_OK;
    if (!d_out)
        return fail(ctx, DST_ERR_ARG, "null output pointer");

    if (measure == DST_TN93 && out_kind == DST_OUT_DISTANCE) {
        int rc = need_counts(ctx, rows, stream);
        if (!rc && &cols != &rows)
            rc = need_counts(ctx, cols, stream);
        if (rc)
            return rc;
    }
    // ---- path: dense bit-planes (work ~ L), consensus-delta (work ~ differences from a reference sequence), or the
    // hybrid of the two (hot columns dense, the rest by lists)
    int rc = DST_OK;
    int path = DST_PATH_DENSE;
    // a launch the dense kernels finish in less time than the lists take to set up goes dense unasked
    const bool tiny = ctx->path == DST_PATH_AUTO && (double)total_pairs * (double)cols.len < ctx->prep_min_work && !cols.ref.valid;
    // a set whose preparation was shared out over the ranks (dst_upload_shared) has the lists of every record but only
    // this rank's planes: the consensus path, whatever the cost model would say
    const bool partial = rows.partial || cols.partial;
    if (partial && (ctx->path == DST_PATH_DENSE || ctx->path == DST_PATH_HYBRID))
        return fail(ctx, DST_ERR_STATE, "a set uploaded with dst_upload_shared runs on the consensus path only");
    if (ctx->path != DST_PATH_DENSE && (!tiny || partial) && consensus_shape_ok(rows, cols)) {
        rc = ensure_lut(ctx);
        if (!rc)
            rc = ensure_ref(ctx, cols, stream);
        if (rc)
            return rc;
        const uint64_t n_panels = (cols.n + kPanelCols - 1) / kPanelCols;
        const uint64_t tiles_est = std::max<uint64_t>(1, n_panels * ((re - rb + kConsensusRowsPerTile - 1) / kConsensusRowsPerTile) / (square ? 2 : 1));
        path = partial ? DST_PATH_CONSENSUS : ctx->path != DST_PATH_AUTO ? ctx->path
                                          : cheapest_path(rows, cols, measure, total_pairs, (uint32_t)std::min<uint64_t>(tiles_est, 0xFFFFFFFFu),
                                                          (double)total_pairs / (double)std::max<uint64_t>(pairs_in_rows(square, cols.n, 0, rows.n), 1));
        const uint64_t n_hot = cols.ref.h_stats[4];
        if (path == DST_PATH_HYBRID && (n_hot == 0 || n_hot * 2 > cols.len))
            path = n_hot == 0 ? DST_PATH_CONSENSUS : DST_PATH_DENSE;  // nothing hot / mostly hot: the plain paths
        if (path != DST_PATH_DENSE) {
            const bool without_hot = path == DST_PATH_HYBRID;
            if (!square && cols.runs.active) {
                // a column set whose run records' lists are stripped serves the square job only (the corrections are
                // between its own records): two files / stream batches rebuild its lists from the planes, whole
                cols.runs.active = false;
                cols.rec.valid = cols.site.valid = cols.rec.pre_valid = false;
                cols.aconst_family = -1;
            }
            rc = ensure_index(ctx, cols, cols, true, without_hot, stream);
            if (!rc && &rows != &cols)
                rc = ensure_index(ctx, rows, cols, false, without_hot, stream);
            if (rc == DST_ERR_CAPACITY && !partial)
                path = DST_PATH_DENSE;  // denser than the lists can index: the dense path handles any input
            else if (rc)
                return rc;
        }
    }
    if (path != DST_PATH_DENSE) {
        const int fam = family_of(measure);
        const bool wide = rows.len >= 65536;
        const bool hybrid = path == DST_PATH_HYBRID;
        rc = ensure_aconst(ctx, cols, cols, fam, wide, stream);
        if (!rc && &rows != &cols)
            rc = ensure_aconst(ctx, rows, cols, fam, wide, stream);
        const void *d_tiles = nullptr;
        uint32_t ntiles = 0;
        // rows per tile: 32, fewer for launches that would not fill the GPU a few times over (256 CUs x 3 workgroups:
        // a 10,000-record job is 800 tiles of 32 rows — one round and a nearly empty second one)
        uint32_t rows_per_tile = kConsensusRowsPerTile;
        {
            const uint64_t n_panels = (cols.n + kPanelCols - 1) / kPanelCols;
            auto tiles_at = [&](uint32_t r) { return n_panels * ((re - rb + r - 1) / r) / (square ? 2 : 1); };
            while (rows_per_tile > 8 && tiles_at(rows_per_tile) < 8 * 768)
                rows_per_tile /= 2;
        }
        if (!rc)
            rc = prepare_schedule(ctx, square, rb, re, cols.n, (int)rows_per_tile, -1, stream, &d_tiles, &ntiles);
        if (rc)
            return rc;
        const void *d_hot = nullptr;
        if (hybrid && ntiles) {
            // the hot columns through the dense kernels: their tallies, in canonical order, into a scratch buffer
            rc = ensure_hot(ctx, cols, cols, stream);
            if (!rc && &rows != &cols)
                rc = ensure_hot(ctx, rows, cols, stream);
            const int hot_kind = wide ? DST_OUT_TALLY : DST_OUT_TALLY16;
            const size_t hot_bytes = dst_out_bytes(measure, hot_kind, total_pairs);
            if (!rc && ctx->hot_tally_bytes < hot_bytes) {
                HIP_TRY(ctx, hipDeviceSynchronize());
                rc = ensure_bytes(ctx, &ctx->hot_tally, &ctx->hot_tally_bytes, hot_bytes);
            }
            if (rc)
                return rc;
            DeviceSet &hrows = *rows.hot, &hcols = *cols.hot;
            const TileShape hts = tile_shape(measure, ctx->variant);
            uint32_t hblocks = 0;
            const BlockDesc *d_hblocks = nullptr;
            rc = prepare_blocks(ctx, square, rb, re, cols.n, hts, stream, &d_hblocks, &hblocks);
            if (rc)
                return rc;
            PairLaunch hp{};
            hp.rows = &hrows;
            hp.cols = &hcols;
            hp.square = square;
            hp.row_begin = rb;
            hp.row_end = re;
            hp.out_base = square ? square_row_start(cols.n, rb) : 0;
            hp.out_kind = hot_kind;
            hp.d_out = ctx->hot_tally;
            hp.d_blocks = d_hblocks;
            hp.nblocks = hblocks;
            hp.ksplit = 1;
            // one scratch buffer per context: a hybrid run on another stream must be done reading it first
            if (ctx->hot_used)
                HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->hot_free, 0));
            if (hblocks)
                HIP_TRY(ctx, launch_pairs(measure, ctx->variant, hp, stream));
            d_hot = ctx->hot_tally;
        }
        // F_k: every (cold) site where the reference is a known base adds f_k(base, base)
        int unit[4];
        site_tallies(measure, 136, 136, unit);
        const uint64_t n_known = cols.ref.h_stats[0] - (hybrid ? cols.ref.h_stats[5] : 0);
        int64_t f[4];
        for (int k = 0; k < 4; ++k)
            f[k] = (int64_t)unit[k] * (int64_t)n_known;
        uint32_t f_words[kMaxWords];
        pack_tallies(fam, wide, f, f_words);
        ConsensusLaunch cl{};
        cl.rows = &rows;
        cl.cols = &cols;
        cl.square = square;
        cl.row_begin = rb;
        cl.row_end = re;
        cl.out_base = square ? square_row_start(cols.n, rb) : 0;
        cl.out_kind = out_kind;
        cl.d_out = d_out;
        cl.d_tiles = static_cast<const ConsensusTile *>(d_tiles);
        cl.ntiles = ntiles;
        cl.wide = wide;
        cl.d_lut = ctx->d_lut;
        cl.d_hot = d_hot;
        {   // the sample's view of this launch's event load (the same figures the path choice reads)
            const uint64_t *st = cols.ref.h_stats;
            const double S = (double)std::max<uint64_t>(st[3], 1);
            const double f = run_scale(cols);
            const double events = f * f * (double)(hybrid ? st[7] : st[2]) / (S * S), list = f * (double)(hybrid ? st[6] : st[1]) / S;
            // (a run record's row or column adds one word per pair from the correction tables: like an event, a cheaper one)
            const double run_adds = cols.runs.active && square ? 2.0 * cols.runs.n_run / std::max<double>((double)cols.n, 1.0) : 0.0;
            // The 4 + 4 wave split beyond event-heavy launches (raw at 50,000 x 30,000 unless noted): any run records at all
            // (1 % of the records: 2.36 against 2.65 ms, 5 %: 3.14 against 3.38 — their adds come at the point of use, not
            // through the events' pipeline); hybrid launches (the event waves also bring in the hot columns' tallies, a
            // word per pair); launches of a few rounds of workgroups (10,000 records: raw 0.121 -> 0.117 ms, n_high 0.111 ->
            // 0.108, k80 0.199 -> 0.193; even at 20,000, 0.4 % slower at 30,000: with little to overlap it with, a tile's
            // first events are what a workgroup waits for).  Not tn93, whose eight waves all take both roles
            // (event_waves()), and not jc69 (0.128 -> 0.154 ms at 10,000 records: its f64 output wants the six waves).
            const bool split_helps = measure != DST_TN93 && measure != DST_JC69 &&
                                     (run_adds > 0.0 || d_hot || total_pairs < 120000000ull);
            cl.heavy_events = events + run_adds > 1.0 ? 2 : events + run_adds > 0.3 || list > 100.0 || split_helps ? 1 : 0;
        }
        ctx->last_path = path;
        if (ntiles) {
            if (int rc_t = timer_begin(ctx, 0, stream))
                return rc_t;
            HIP_TRY(ctx, launch_consensus_pairs(measure, cl, f_words, stream));
            if (int rc_t = timer_end(ctx, 0, stream))
                return rc_t;
            if (d_hot) {
                HIP_TRY(ctx, hipEventRecord(ctx->hot_free, stream));
                ctx->hot_used = true;
            }
            rc = note_run(ctx, stream);  // a rebuild of the lists on another stream waits for this run
            if (rc)
                return rc;
        }
        if (!stream_v)
            HIP_TRY(ctx, hipStreamSynchronize(stream));
        return DST_OK;
    }
    ctx->last_path = DST_PATH_DENSE;
    rc = ensure_derived(ctx, cols, stream);
    if (!rc && &rows != &cols)
        rc = ensure_derived(ctx, rows, stream);
    if (rc)
        return rc;
    const TileShape ts = tile_shape(measure, ctx->variant);
    uint32_t nblocks = 0;
    const BlockDesc *d_blocks = nullptr;
    rc = prepare_blocks(ctx, square, rb, re, cols.n, ts, stream, &d_blocks, &nblocks);
    if (rc)
        return rc;
    PairLaunch pl{};
    pl.rows = &rows;
    pl.cols = &cols;
    pl.square = square;
    pl.row_begin = rb;
    pl.row_end = re;
    pl.out_base = square ? square_row_start(cols.n, rb) : 0;
    pl.out_kind = out_kind;
    pl.d_out = d_out;
    pl.d_blocks = d_blocks;
    pl.nblocks = nblocks;
    // Few tiles but a long alignment (small sets, streamed batches): split the sweep over L so the
    // launch still fills the 256 CUs; partial tallies are combined with integer atomics (exact).
    uint32_t ksplit = 1;
    if (ctx->ksplit >= 1) {
        ksplit = (uint32_t)ctx->ksplit;
    } else if (nblocks && nblocks < 1024 && rows.nchunks >= 16 && out_kind != DST_OUT_TALLY16) {
        ksplit = (uint32_t)std::min<uint64_t>({(2048 + nblocks - 1) / nblocks, rows.nchunks / 8, (uint64_t)64});
    }
    ksplit = (uint32_t)std::min<uint64_t>(std::max<uint32_t>(ksplit, 1), std::max<size_t>(rows.nchunks, 1));
    if (out_kind == DST_OUT_TALLY16)
        ksplit = 1;  // no 16-bit atomics: such launches sweep L in one piece
    pl.ksplit = ksplit;
    const bool f64_out = out_kind == DST_OUT_DISTANCE && !measure_is_int(measure);
    if (nblocks && ksplit > 1) {
        if (f64_out) {
            const size_t want = (size_t)total_pairs * tally_width(measure) * sizeof(uint32_t);
            if (ctx->scratch_bytes < want) {
                HIP_TRY(ctx, hipDeviceSynchronize());  // an earlier run (any stream) may still read the old one
                rc = ensure_bytes(ctx, (void **)&ctx->scratch, &ctx->scratch_bytes, want);
                if (rc)
                    return rc;
            }
            // one meeting buffer per context: a run on another stream must be done with it first
            if (ctx->scratch_used)
                HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->scratch_free, 0));
            HIP_TRY(ctx, hipMemsetAsync(ctx->scratch, 0, want, stream));
            pl.out_kind = DST_OUT_TALLY;
            pl.d_out = ctx->scratch;
        } else {
            HIP_TRY(ctx, hipMemsetAsync(d_out, 0, need, stream));
        }
    }
    if (nblocks) {
        if (int rc_t = timer_begin(ctx, 0, stream))
            return rc_t;
        HIP_TRY(ctx, launch_pairs(measure, ctx->variant, pl, stream));
        if (int rc_t = timer_end(ctx, 0, stream))
            return rc_t;
        if (ksplit > 1 && f64_out) {
            HIP_TRY(ctx, launch_finalize(measure, pl, ctx->scratch, false, d_out, stream));
            HIP_TRY(ctx, hipEventRecord(ctx->scratch_free, stream));
            ctx->scratch_used = true;
        }
    }
    if (!stream_v)
        HIP_TRY(ctx, hipStreamSynchronize(stream));
    return DST_OK;
}

int run_host(dst_ctx *ctx, int measure, bool square, int row_slot, int col_slot, uint64_t rb,
             uint64_t re, int out_kind, void *h_out, size_t cap)
{
    if (!ctx)
        return DST_ERR_ARG;
    if (row_slot < 0 || row_slot > 1 || col_slot < 0 || col_slot > 1)
        return fail(ctx, DST_ERR_ARG, "slot must be 0 or 1");
    const DeviceSet &cols = ctx->set[col_slot];
    const uint64_t pairs = pairs_in_rows(square, cols.n, rb, re);
    const size_t bytes = dst_out_bytes(measure, out_kind, pairs);
    if (bytes > cap)
        return fail(ctx, DST_ERR_CAPACITY, "output buffer too small for the requested rows");
    if (bytes == 0)
        return run_common(ctx, measure, square, row_slot, col_slot, rb, re, out_kind, nullptr, 0, nullptr);
    if (!h_out)
        return fail(ctx, DST_ERR_ARG, "null output pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // one grow-only device buffer per context for the host-buffer forms (no allocation per call)
    int rc = ensure_bytes(ctx, &ctx->host_out, &ctx->host_out_bytes, bytes);
    if (rc)
        return rc;
    rc = run_common(ctx, measure, square, row_slot, col_slot, rb, re, out_kind, ctx->host_out, bytes, (void *)ctx->stream);
    if (rc == DST_OK) {
        hipError_t e = hipMemcpyAsync(h_out, ctx->host_out, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess)
            rc = fail_hip(ctx, e, "hipMemcpy(D2H results)");
    }
    return rc;
}

}  // namespace dst

extern "C" {

int dst_device_count(int *count)
{
    if (!count)
        return DST_ERR_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return DST_ERR_HIP;
    }
    *count = n;
    return DST_OK;
}

const char *dst_last_error(const dst_ctx *ctx)
{
    if (ctx)
        return ctx->err.c_str();
    std::lock_guard<std::mutex> lk(g_create_mu);
    return g_create_err.c_str();
}

int dst_create(int device, dst_ctx **out)
{
    if (!out)
        return DST_ERR_ARG;
    *out = nullptr;
    auto bail = [&](dst_ctx *c, hipError_t e, const char *what) {
        std::lock_guard<std::mutex> lk(g_create_mu);
        g_create_err = std::string(what) + ": " + hipGetErrorString(e);
        delete c;
        return e == hipErrorOutOfMemory ? DST_ERR_NOMEM : DST_ERR_HIP;
    };
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return bail(nullptr, e != hipSuccess ? e : hipErrorNoDevice,
                    "no HIP device (libdistance_hip has no CPU path)");
    if (device < 0 || device >= count) {
        std::lock_guard<std::mutex> lk(g_create_mu);
        g_create_err = "device index out of range";
        return DST_ERR_ARG;
    }
    dst_ctx *c = new (std::nothrow) dst_ctx;
    if (!c)
        return DST_ERR_NOMEM;
    c->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess)
        return bail(c, e, "hipSetDevice");
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess)
        return bail(c, e, "hipStreamCreate");
    for (auto &t : c->timer)
        for (int k = 0; k < dst_ctx::kTimerRing; ++k)
            if ((e = hipEventCreate(&t.begin[k])) != hipSuccess || (e = hipEventCreate(&t.end[k])) != hipSuccess)
                return bail(c, e, "hipEventCreate");
    if ((e = hipMalloc((void **)&c->d_first_bad, sizeof(unsigned long long))) != hipSuccess)
        return bail(c, e, "hipMalloc");
    if ((e = hipHostMalloc((void **)&c->h_report, 16 * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess)
        return bail(c, e, "hipHostMalloc");
    if ((e = hipHostGetDevicePointer((void **)&c->d_report, c->h_report, 0)) != hipSuccess)
        return bail(c, e, "hipHostGetDevicePointer");
    if ((e = hipEventCreateWithFlags(&c->scratch_free, hipEventDisableTiming)) != hipSuccess)
        return bail(c, e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->hot_free, hipEventDisableTiming)) != hipSuccess)
        return bail(c, e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->prep_event, hipEventDisableTiming)) != hipSuccess)
        return bail(c, e, "hipEventCreate");
    for (auto &r : c->recent)
        if ((e = hipEventCreateWithFlags(&r.event, hipEventDisableTiming)) != hipSuccess)
            return bail(c, e, "hipEventCreate");
    *out = c;
    return DST_OK;
}

int dst_destroy(dst_ctx *ctx)
{
    if (!ctx)
        return DST_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    free_set(ctx->set[0]);
    free_set(ctx->set[1]);
    if (ctx->stage)
        (void)hipFree(ctx->stage);
    for (auto &s : ctx->schedules)
        if (s.d_blocks)
            (void)hipFree(s.d_blocks);
    for (auto &sh : ctx->shared)
        for (void *b : {sh.send, sh.recv, (void *)sh.off_local})
            if (b)
                (void)hipFree(b);
    for (void *b : {(void *)ctx->d_lut, (void *)ctx->d_total, (void *)ctx->scan_tmp, ctx->host_out, ctx->hot_tally, ctx->text_res,
                    ctx->text_num, (void *)ctx->text_len, (void *)ctx->text_scan, (void *)ctx->text_buf, (void *)ctx->text_flag, ctx->text_ties,
                    (void *)ctx->ids[0].off, (void *)ctx->ids[0].chars, (void *)ctx->ids[1].off, (void *)ctx->ids[1].chars})
        if (b)
            (void)hipFree(b);
    if (ctx->scratch)
        (void)hipFree(ctx->scratch);
    if (ctx->text_ties_host)
        (void)hipHostFree(ctx->text_ties_host);
    if (ctx->scratch_free)
        (void)hipEventDestroy(ctx->scratch_free);
    if (ctx->hot_free)
        (void)hipEventDestroy(ctx->hot_free);
    if (ctx->prep_event)
        (void)hipEventDestroy(ctx->prep_event);
    for (auto &r : ctx->recent)
        if (r.event)
            (void)hipEventDestroy(r.event);
    if (ctx->d_first_bad)
        (void)hipFree(ctx->d_first_bad);
    if (ctx->h_report)
        (void)hipHostFree(ctx->h_report);
    for (auto &t : ctx->timer)
        for (int k = 0; k < dst_ctx::kTimerRing; ++k) {
            if (t.begin[k])
                (void)hipEventDestroy(t.begin[k]);
            if (t.end[k])
                (void)hipEventDestroy(t.end[k]);
        }
    if (ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return DST_OK;
}

int dst_variant_count(int measure) { return variant_count(measure); }

const char *dst_build_flags(void)
{
    const char *f = consensus_build_flags();
    return *f == ' ' ? f + 1 : f;
}

int dst_set_ksplit(dst_ctx *ctx, int ksplit)
{
    if (!ctx || ksplit < 0)
        return DST_ERR_ARG;
    ctx->ksplit = ksplit;
    return DST_OK;
}

int dst_set_prep_threshold(dst_ctx *ctx, double site_comparisons)
{
    if (!ctx)
        return DST_ERR_ARG;
    if (!(site_comparisons >= 0.0))
        return fail(ctx, DST_ERR_ARG, "threshold must be >= 0");
    ctx->prep_min_work = site_comparisons;
    return DST_OK;
}

int dst_set_path(dst_ctx *ctx, int path)
{
    if (!ctx || path < DST_PATH_AUTO || path > DST_PATH_HYBRID)
        return DST_ERR_ARG;
    ctx->path = path;
    return DST_OK;
}

int dst_last_path(const dst_ctx *ctx) { return ctx ? ctx->last_path : -1; }

int dst_run_records(const dst_ctx *ctx, int slot, uint64_t *run_records, uint64_t *entries_removed)
{
    if (!ctx || slot < 0 || slot > 1)
        return DST_ERR_ARG;
    const DeviceSet &s = ctx->set[slot];
    if (run_records)
        *run_records = s.runs.active ? s.runs.n_run : 0;
    if (entries_removed)
        *entries_removed = s.runs.active ? s.runs.removed : 0;
    return DST_OK;
}

int dst_planes_stored(const dst_ctx *ctx, int slot, int *stored)
{
    if (!ctx || slot < 0 || slot > 1 || !stored)
        return DST_ERR_ARG;
    *stored = ctx->set[slot].loaded && !ctx->set[slot].planes_deferred ? 1 : 0;
    return DST_OK;
}

int dst_set_variant(dst_ctx *ctx, int variant)
{
    if (!ctx || variant < 0)
        return DST_ERR_ARG;
    ctx->variant = variant;
    return DST_OK;
}

int dst_upload(dst_ctx *ctx, int slot, const uint8_t *codes, size_t n, size_t len, size_t row_stride,
               const uint32_t *base_counts)
{
    if (!ctx)
        return DST_ERR_ARG;
    if (slot < 0 || slot > 1)
        return fail(ctx, DST_ERR_ARG, "slot must be 0 or 1");
    if (n == 0)
        return fail(ctx, DST_ERR_ARG, "Empty FASTA file");  // src/fastaio.rs:97-99
    if ((len && !codes) || row_stride < len)
        return fail(ctx, DST_ERR_ARG, "null codes or row_stride < len");
    if (n >= 0xFFFFFE00ull || len >= 0xFFFFFF00ull)
        return fail(ctx, DST_ERR_ARG, "n and len must fit 32 bits");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // pitched staging copy: rows land 128-byte aligned whatever the caller's stride
    const size_t pitch = ((len + 127) / 128) * 128;
    const size_t want = std::max<size_t>(pitch * n, 128) + (base_counts ? n * 16 : 0);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int rc = ensure_bytes(ctx, (void **)&ctx->stage, &ctx->stage_bytes, want);
    if (rc)
        return rc;
    if (len)
        HIP_TRY(ctx, hipMemcpy2DAsync(ctx->stage, pitch, codes, row_stride, len, n, hipMemcpyHostToDevice,
                                      ctx->stream));
    uint32_t *d_counts = nullptr;
    if (base_counts) {
        d_counts = reinterpret_cast<uint32_t *>(ctx->stage + std::max<size_t>(pitch * n, 128));
        HIP_TRY(ctx, hipMemcpyAsync(d_counts, base_counts, n * 16, hipMemcpyHostToDevice, ctx->stream));
    }
    rc = pack_from_device(ctx, slot, ctx->stage, n, len, pitch, d_counts, ctx->stream);
    // the bytes are not needed once they are packed: a large staging buffer (a whole loaded set) is given back, a
    // small one (streamed batches) is kept for the next upload
    if (ctx->stage_bytes > ((size_t)256 << 20)) {
        (void)hipFree(ctx->stage);
        ctx->stage = nullptr;
        ctx->stage_bytes = 0;
    }
    return rc;
}

int dst_upload_device(dst_ctx *ctx, int slot, const void *d_codes, size_t n, size_t len,
                      size_t row_stride, const uint32_t *d_base_counts, void *stream_v)
{
    if (!ctx)
        return DST_ERR_ARG;
    if (slot < 0 || slot > 1)
        return fail(ctx, DST_ERR_ARG, "slot must be 0 or 1");
    if (n == 0)
        return fail(ctx, DST_ERR_ARG, "Empty FASTA file");
    if ((len && !d_codes) || row_stride < len)
        return fail(ctx, DST_ERR_ARG, "null codes or row_stride < len");
    if (n >= 0xFFFFFE00ull || len >= 0xFFFFFF00ull)
        return fail(ctx, DST_ERR_ARG, "n and len must fit 32 bits");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_v ? (hipStream_t)stream_v : ctx->stream;
    return pack_from_device(ctx, slot, (const uint8_t *)d_codes, n, len, row_stride, d_base_counts, stream);
}

int dst_set_info(const dst_ctx *ctx, int slot, size_t *n, size_t *len)
{
    if (!ctx || slot < 0 || slot > 1)
        return DST_ERR_ARG;
    if (!ctx->set[slot].loaded)
        return DST_ERR_STATE;
    if (n)
        *n = ctx->set[slot].n;
    if (len)
        *len = ctx->set[slot].len;
    return DST_OK;
}

int dst_get_base_counts(dst_ctx *ctx, int slot, uint32_t *counts)
{
    if (!ctx || slot < 0 || slot > 1 || !counts)
        return DST_ERR_ARG;
    DeviceSet &s = ctx->set[slot];
    if (!s.loaded)
        return fail(ctx, DST_ERR_STATE, "set not uploaded");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = need_counts(ctx, s, ctx->stream);
    if (rc)
        return rc;
    HIP_TRY(ctx, hipMemcpyAsync(counts, s.counts, s.n * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DST_OK;
}

int dst_consensus(dst_ctx *ctx, int both_slots, uint8_t *cons, size_t cap)
{
    if (!ctx || !cons)
        return DST_ERR_ARG;
    DeviceSet &a = ctx->set[0];
    if (!a.loaded)
        return fail(ctx, DST_ERR_STATE, "set not uploaded");
    const bool two = both_slots && ctx->set[1].loaded;
    if (a.partial || (two && ctx->set[1].partial))
        return fail(ctx, DST_ERR_STATE, "not available for a set uploaded with dst_upload_shared (this rank holds the planes of its own records only)");
    if (two && ctx->set[1].len != a.len)
        return fail(ctx, DST_ERR_STATE, "Different length sequences in alignment(s)");
    if (cap < a.len)
        return fail(ctx, DST_ERR_CAPACITY, "consensus buffer shorter than the alignment");
    if (a.len == 0)
        return DST_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t *d_hist = nullptr;
    const size_t bytes = a.len * 3 * sizeof(uint32_t);
    HIP_TRY(ctx, hipMalloc((void **)&d_hist, bytes));
    std::vector<uint32_t> hist(a.len * 3);
    int rc_p = ensure_planes(ctx, a, ctx->stream);
    if (!rc_p && two)
        rc_p = ensure_planes(ctx, ctx->set[1], ctx->stream);
    if (rc_p) {
        (void)hipFree(d_hist);
        return rc_p;
    }
    hipError_t e = hipMemsetAsync(d_hist, 0, bytes, ctx->stream);
    if (e == hipSuccess)
        e = launch_site_hist(a, d_hist, ctx->stream);
    if (e == hipSuccess && two)
        e = launch_site_hist(ctx->set[1], d_hist, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(hist.data(), d_hist, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_hist);
    if (e != hipSuccess)
        return fail_hip(ctx, e, "consensus counts");
    const uint64_t records = a.n + (two ? ctx->set[1].n : 0);
    static const uint8_t back_translate[4] = {136, 72, 40, 24};  // [A, G, C, T], src/fastaio.rs:314-316
    for (size_t i = 0; i < a.len; ++i) {
        const uint64_t g = hist[3 * i], c = hist[3 * i + 1], t = hist[3 * i + 2];
        const uint64_t counts[4] = {records - g - c - t, g, c, t};  // every non-G/C/T byte is looked up as A
        size_t maxidx = 0;
        uint64_t maxval = 0;
        for (size_t k = 0; k < 4; ++k)
            if (counts[k] > maxval) {  // strict: ties keep the earlier base (src/fastaio.rs:322-327)
                maxval = counts[k];
                maxidx = k;
            }
        cons[i] = back_translate[maxidx];
    }
    return DST_OK;
}

int dst_differences(dst_ctx *ctx, int slot, const uint8_t *other, size_t len, uint64_t *offsets, uint32_t *sites,
                    size_t cap_sites, uint64_t *total_out)
{
    if (!ctx || slot < 0 || slot > 1 || !offsets || !total_out || (len && !other))
        return DST_ERR_ARG;
    DeviceSet &s = ctx->set[slot];
    if (!s.loaded)
        return fail(ctx, DST_ERR_STATE, "set not uploaded");
    if (s.partial)
        return fail(ctx, DST_ERR_STATE, "not available for a set uploaded with dst_upload_shared (this rank holds the planes of its own records only)");
    if (len >= kSiteMask)   // the list entries carry the site in kSiteBits bits
        return fail(ctx, DST_ERR_CAPACITY, "alignments of 2^25 sites or more are beyond dst_differences");
    if (len != s.len) {
        char msg[128];
        std::snprintf(msg, sizeof msg, "Different length sequences in alignment(s): %zu vs %zu", len, s.len);
        return fail(ctx, DST_ERR_STATE, msg);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_lut(ctx);
    if (!rc)
        rc = ensure_planes(ctx, s, ctx->stream);   // (the lists against `other` are taken from the planes)
    if (rc)
        return rc;
    // `other` as A, G, C, T planes (sites past len are N on both sides: never a difference)
    std::vector<uint32_t> planes(4 * s.nchunks * 4, 0xFFFFFFFFu);
    for (size_t i = 0; i < len; ++i)
        for (int p = 0; p < 4; ++p)
            if (!((other[i] >> (7 - p)) & 1u))
                planes[(p * s.nchunks + i / kChunkSites) * 4 + (i % kChunkSites) / 32] &= ~(1u << (i % 32));
    uint4 *d_ref = nullptr;
    uint32_t *d_off = nullptr, *d_ent = nullptr, *d_tmp = nullptr;
    std::vector<uint32_t> off32(s.n + 1);
    unsigned long long total = 0;
    hipError_t e = hipMalloc((void **)&d_ref, planes.size() * sizeof(uint32_t));
    if (e == hipSuccess)
        e = hipMalloc((void **)&d_off, (s.n + 1) * sizeof(uint32_t));
    if (e == hipSuccess)
        e = hipMalloc((void **)&d_tmp, scan_tmp_words(s.n + 1) * sizeof(uint32_t));
    auto done = [&](int status) {
        for (void *b : {(void *)d_ref, (void *)d_off, (void *)d_ent, (void *)d_tmp})
            if (b)
                (void)hipFree(b);
        return status;
    };
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_ref, planes.data(), planes.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemsetAsync(d_off, 0, (s.n + 1) * sizeof(uint32_t), ctx->stream);
    if (e == hipSuccess)
        e = hipMemsetAsync(ctx->d_total, 0, sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess)
        e = launch_index(s, d_ref, nullptr, false, true, d_off, nullptr, ctx->d_total,
                         ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(&total, ctx->d_total, sizeof total, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess)
        return done(fail_hip(ctx, e, "differences (count)"));
    if (total > kMaxListEntries)
        return done(fail(ctx, DST_ERR_CAPACITY, "more than 2^31 differences"));
    e = launch_exclusive_scan(d_off, s.n + 1, d_tmp, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(off32.data(), d_off, (s.n + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess)
        return done(fail_hip(ctx, e, "differences (scan)"));
    for (size_t r = 0; r <= s.n; ++r)
        offsets[r] = off32[r];
    *total_out = total;
    if (!sites)
        return done(DST_OK);
    if (cap_sites < total)
        return done(fail(ctx, DST_ERR_CAPACITY, "sites buffer too small"));
    if (total == 0)
        return done(DST_OK);
    e = hipMalloc((void **)&d_ent, total * sizeof(uint32_t));
    if (e == hipSuccess)
        e = launch_index(s, d_ref, nullptr, true, true, d_off, d_ent, ctx->d_total,
                         ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(sites, d_ent, total * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess)
        return done(fail_hip(ctx, e, "differences (fill)"));
    for (uint64_t k = 0; k < total; ++k)
        sites[k] &= kSiteMask;  // drop the reference class and the nibble the pair kernel's lists carry
    return done(DST_OK);
}

int dst_run_square(dst_ctx *ctx, int measure, uint64_t row_begin, uint64_t row_end, int out_kind,
                   void *d_out, size_t cap, void *stream)
{
    return run_common(ctx, measure, true, 0, 0, row_begin, row_end, out_kind, d_out, cap, stream);
}

int dst_run_rect(dst_ctx *ctx, int measure, int row_slot, int col_slot, uint64_t row_begin,
                 uint64_t row_end, int out_kind, void *d_out, size_t cap, void *stream)
{
    return run_common(ctx, measure, false, row_slot, col_slot, row_begin, row_end, out_kind, d_out, cap,
                      stream);
}

int dst_finalize_device(dst_ctx *ctx, int measure, int square, int row_slot, int col_slot, uint64_t row_begin,
                        uint64_t row_end, int tally_kind, const void *d_tallies, void *d_out, size_t cap,
                        void *stream_v)
{
    if (!ctx)
        return DST_ERR_ARG;
    if (measure < DST_N || measure > DST_TN93)
        return fail(ctx, DST_ERR_ARG, "unknown measure");
    const bool close = (tally_kind & DST_FIN_CLOSE) != 0;
    tally_kind &= ~DST_FIN_CLOSE;
    if (tally_kind != DST_OUT_TALLY && tally_kind != DST_OUT_TALLY16)
        return fail(ctx, DST_ERR_ARG, "tally_kind must be DST_OUT_TALLY or DST_OUT_TALLY16");
    if (row_slot < 0 || row_slot > 1 || col_slot < 0 || col_slot > 1)
        return fail(ctx, DST_ERR_ARG, "slot must be 0 or 1");
    DeviceSet &rows = ctx->set[row_slot];
    DeviceSet &cols = ctx->set[col_slot];
    if (!rows.loaded || !cols.loaded)
        return fail(ctx, DST_ERR_STATE, "set not uploaded");
    if (row_begin > row_end || row_end > rows.n)
        return fail(ctx, DST_ERR_ARG, "row range out of bounds");
    const uint64_t pairs = pairs_in_rows(square != 0, cols.n, row_begin, row_end);
    if (pairs * 8 > cap)
        return fail(ctx, DST_ERR_CAPACITY, "output buffer too small for the requested rows");
    if (pairs == 0)
        return DST_OK;
    if (!d_tallies || !d_out)
        return fail(ctx, DST_ERR_ARG, "null pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_v ? (hipStream_t)stream_v : ctx->stream;
    if (measure == DST_TN93) {
        int rc = need_counts(ctx, rows, stream);
        if (!rc && &cols != &rows)
            rc = need_counts(ctx, cols, stream);
        if (rc)
            return rc;
    }
    PairLaunch pl{};
    pl.rows = &rows;
    pl.cols = &cols;
    pl.square = square != 0;
    pl.row_begin = row_begin;
    pl.row_end = row_end;
    pl.out_base = square ? square_row_start(cols.n, row_begin) : 0;
    HIP_TRY(ctx, launch_finalize(measure, pl, d_tallies, tally_kind == DST_OUT_TALLY16, d_out, stream, close));
    if (!stream_v)
        HIP_TRY(ctx, hipStreamSynchronize(stream));
    return DST_OK;
}

int dst_host_alloc(size_t bytes, void **ptr)
{
    if (!ptr)
        return DST_ERR_ARG;
    *ptr = nullptr;
    const hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault);
    return e == hipSuccess ? DST_OK : (e == hipErrorOutOfMemory ? DST_ERR_NOMEM : DST_ERR_HIP);
}

int dst_host_free(void *ptr)
{
    if (!ptr)
        return DST_OK;
    return hipHostFree(ptr) == hipSuccess ? DST_OK : DST_ERR_HIP;
}

int dst_run_slabs(dst_ctx *ctx, int measure, int square, int row_slot, int col_slot, int out_kind,
                  uint64_t max_pairs, dst_slab_sink sink, void *user)
{
    if (!ctx)
        return DST_ERR_ARG;
    if (!sink || max_pairs == 0)
        return fail(ctx, DST_ERR_ARG, "null sink or max_pairs == 0");
    if (square) {
        row_slot = 0;
        col_slot = 0;
    }
    if (row_slot < 0 || row_slot > 1 || col_slot < 0 || col_slot > 1)
        return fail(ctx, DST_ERR_ARG, "slot must be 0 or 1");
    if (!ctx->set[row_slot].loaded || !ctx->set[col_slot].loaded)
        return fail(ctx, DST_ERR_STATE, "set not uploaded");
    const uint64_t n_rows = ctx->set[row_slot].n, n_cols = ctx->set[col_slot].n;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // two device + two pinned host buffers sized for the largest slab
    struct Slab {
        uint64_t rb, re, first, pairs;
    };
    std::vector<Slab> slabs;
    uint64_t first = 0, biggest = 0;
    for (uint64_t rb = 0; rb < n_rows;) {
        uint64_t re = rb, pairs = 0;
        while (re < n_rows) {
            const uint64_t row_pairs = square ? (n_cols - re - 1) : n_cols;
            if (re > rb && pairs + row_pairs > max_pairs)
                break;
            pairs += row_pairs;
            ++re;
        }
        if (pairs)
            slabs.push_back({rb, re, first, pairs});
        first += pairs;
        biggest = std::max(biggest, pairs);
        rb = re;
    }
    if (slabs.empty())
        return DST_OK;
    const size_t bytes = dst_out_bytes(measure, out_kind, biggest);
    void *d_buf[2] = {nullptr, nullptr}, *h_buf[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    int rc = DST_OK;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(ctx->stream);
        for (int k = 0; k < 2; ++k) {
            if (d_buf[k])
                (void)hipFree(d_buf[k]);
            if (h_buf[k])
                (void)hipHostFree(h_buf[k]);
            if (done[k])
                (void)hipEventDestroy(done[k]);
        }
    };
#define SLAB_TRY(call)                                   \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) {                          \
            rc = fail_hip(ctx, e_, #call);               \
            cleanup();                                   \
            return rc;                                   \
        }                                                \
    } while (0)
    for (int k = 0; k < 2; ++k) {
        SLAB_TRY(hipMalloc(&d_buf[k], bytes));
        SLAB_TRY(hipHostMalloc(&h_buf[k], bytes, hipHostMallocDefault));
        SLAB_TRY(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
    }
    auto issue = [&](size_t k) -> int {  // compute slab k and start its copy back, all on ctx->stream
        const Slab &s = slabs[k];
        const size_t nb = dst_out_bytes(measure, out_kind, s.pairs);
        int r = run_common(ctx, measure, square != 0, row_slot, col_slot, s.rb, s.re, out_kind, d_buf[k & 1], nb,
                           (void *)ctx->stream);
        if (r)
            return r;
        hipError_t e = hipMemcpyAsync(h_buf[k & 1], d_buf[k & 1], nb, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess)
            e = hipEventRecord(done[k & 1], ctx->stream);
        return e == hipSuccess ? DST_OK : fail_hip(ctx, e, "copy back");
    };
    rc = issue(0);
    for (size_t k = 0; rc == DST_OK && k < slabs.size(); ++k) {
        if (k + 1 < slabs.size())
            rc = issue(k + 1);  // the other buffer pair: slab k-1's sink call has returned
        if (rc)
            break;
        SLAB_TRY(hipEventSynchronize(done[k & 1]));
        if (sink(user, slabs[k].first, slabs[k].pairs, slabs[k].rb, slabs[k].re, h_buf[k & 1]) != 0)
            rc = fail(ctx, DST_ERR_STATE, "stopped by sink");
    }
#undef SLAB_TRY
    cleanup();
    return rc;
}

int dst_run_square_host(dst_ctx *ctx, int measure, uint64_t row_begin, uint64_t row_end, int out_kind,
                        void *h_out, size_t cap)
{
    return run_host(ctx, measure, true, 0, 0, row_begin, row_end, out_kind, h_out, cap);
}

int dst_run_rect_host(dst_ctx *ctx, int measure, int row_slot, int col_slot, uint64_t row_begin,
                      uint64_t row_end, int out_kind, void *h_out, size_t cap)
{
    return run_host(ctx, measure, false, row_slot, col_slot, row_begin, row_end, out_kind, h_out, cap);
}

int dst_plan_tiles(int square, uint64_t row_begin, uint64_t row_end, uint64_t n_cols, int measure,
                   int variant, uint32_t *ij, size_t cap_tiles, size_t *count, int *tile_rows,
                   int *tile_cols)
{
    if (measure < DST_N || measure > DST_TN93 || !count)
        return DST_ERR_ARG;
    const TileShape ts = tile_shape(measure, variant);
    if (tile_rows)
        *tile_rows = ts.bm;
    if (tile_cols)
        *tile_cols = ts.bn;
    const std::vector<BlockDesc> blocks = build_blocks(square != 0, row_begin, row_end, n_cols, ts);
    *count = blocks.size();
    if (!ij)
        return DST_OK;
    if (cap_tiles < blocks.size())
        return DST_ERR_CAPACITY;
    for (size_t k = 0; k < blocks.size(); ++k) {
        ij[2 * k] = blocks[k].i0;
        ij[2 * k + 1] = blocks[k].j0;
    }
    return DST_OK;
}

int dst_last_kernel_ms(dst_ctx *ctx, float *pair_ms, float *finalize_ms, float *pack_ms)
{
    if (!ctx)
        return DST_ERR_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->timed_pair) {
        const int k = (int)((ctx->timer[0].seq - 1) % dst_ctx::kTimerRing);
        HIP_TRY(ctx, hipEventSynchronize(ctx->timer[0].end[k]));
        HIP_TRY(ctx, hipEventElapsedTime(&ctx->pair_ms, ctx->timer[0].begin[k], ctx->timer[0].end[k]));
    }
    if (ctx->timed_pack) {
        const int k = (int)((ctx->timer[1].seq - 1) % dst_ctx::kTimerRing);
        HIP_TRY(ctx, hipEventSynchronize(ctx->timer[1].end[k]));
        HIP_TRY(ctx, hipEventElapsedTime(&ctx->pack_ms, ctx->timer[1].begin[k], ctx->timer[1].end[k]));
    }
    if (pair_ms)
        *pair_ms = ctx->pair_ms;
    if (finalize_ms)
        *finalize_ms = 0.0f;  // finalisation is fused into the pair kernel's epilogue
    if (pack_ms)
        *pack_ms = ctx->pack_ms;
    return DST_OK;
}

int dst_kernel_ms_mean(dst_ctx *ctx, int reset, float *pair_ms, int *pair_launches, float *pack_ms, int *pack_launches)
{
    if (!ctx)
        return DST_ERR_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float mean[2] = {0, 0};
    int count[2] = {0, 0};
    for (int w = 0; w < 2; ++w) {
        dst_ctx::Timer &t = ctx->timer[w];
        const uint64_t first = std::max(t.mark, t.seq > (uint64_t)dst_ctx::kTimerRing ? t.seq - dst_ctx::kTimerRing : 0);
        double sum = 0;
        for (uint64_t q = first; q < t.seq; ++q) {
            const int k = (int)(q % dst_ctx::kTimerRing);
            float ms = 0;
            HIP_TRY(ctx, hipEventSynchronize(t.end[k]));
            HIP_TRY(ctx, hipEventElapsedTime(&ms, t.begin[k], t.end[k]));
            sum += ms;
        }
        count[w] = (int)(t.seq - first);
        mean[w] = count[w] ? (float)(sum / count[w]) : 0.0f;
        if (reset)
            t.mark = t.seq;
    }
    if (pair_ms)
        *pair_ms = mean[0];
    if (pair_launches)
        *pair_launches = count[0];
    if (pack_ms)
        *pack_ms = mean[1];
    if (pack_launches)
        *pack_launches = count[1];
    return DST_OK;
}

}  // extern "C"
