// dst_shared.cpp — dst_upload_shared: the preparation of a loaded set shared out over the ranks of a communicator.
//
// Why: one process per GPU, the pair space cut into row ranges (dst_partition_square).  Left alone every rank packs and
// indexes the whole set before it computes its 1/world of the pairs, and for a 50,000 x 30,000 alignment that replicated
// preparation (0.8 ms) is 27 % of a ONE-GPU step — at 8 GPUs it is three times the pair kernel's share.  Here rank k
//   1. samples the reference sequence from the byte matrix (the same 512 records on every rank: the same sequence),
//   2. packs and lists only ITS records [k rmax, (k+1) rmax) — the two passes over the big data,
//   3. puts their list lengths, base counts and entries into a block of fixed size,
// then ONE all-gather (RCCL, or the communicator's own transport) brings every block to every rank, and each rank splices
// the lists of all records into the CSR its pair kernel walks; the site buckets and the per-record constants are built
// from the lists (1/250 of the planes' bytes) locally, as always.  The exchanged volume is the lists: ~13 MB for 50,000
// SARS-CoV-2-like records.  Replaces the worker pool's shared read-only `loaded_fastas` (src/lib.rs:413-458): there every
// worker sees the one copy the main thread prepared (src/lib.rs:219-242); here every GPU prepares a share of it.
//
// The block's entry capacity is the same on every rank without talking: it is derived from the totals of the PREVIOUS
// shared upload (every rank sees every block's header), or a default at first.  Lists that do not fit, a set too diverse
// for lists, a context forced onto the dense kernels: every rank sees that in the same headers / statistics and falls
// back, together, to the replicated upload (dst_upload_device).
#include <algorithm>
#include <cstdio>

#include "dst_ctx.h"

using namespace dst;

namespace dst {
dst_ctx *comm_ctx(dst_comm *c);
int comm_rank(const dst_comm *c);
int comm_world(const dst_comm *c);
int comm_allgather(dst_comm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t stream);
int pack_from_device(dst_ctx *ctx, int slot, const uint8_t *d_codes, size_t n, size_t len, size_t row_stride,
                     const uint32_t *d_counts, hipStream_t stream);
int shape_set(dst_ctx *ctx, DeviceSet &s, size_t n, size_t len);
int alloc_ref(dst_ctx *ctx, DeviceSet &s);
int ensure_lut(dst_ctx *ctx);
int wait_for_other_runs(dst_ctx *ctx, hipStream_t stream);
int publish_prep(dst_ctx *ctx, hipStream_t stream);
}  // namespace dst

extern "C" int dst_upload_shared(dst_comm *comm, int slot, const void *d_codes_v, size_t n, size_t len, size_t row_stride,
                                 int with_counts, void *stream_v)
{
    if (!comm)
        return DST_ERR_ARG;
    dst_ctx *ctx = comm_ctx(comm);
    const int rank = comm_rank(comm), world = comm_world(comm);
    const uint8_t *d_codes = static_cast<const uint8_t *>(d_codes_v);
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
    // shapes and settings the lists cannot serve (the same test on every rank): the replicated upload
    const bool eligible = (ctx->path == DST_PATH_AUTO || ctx->path == DST_PATH_CONSENSUS) && n >= 2 && len > 0 &&
                          n < kEntryMask && len < kSiteMask;
    if (!eligible)
        return pack_from_device(ctx, slot, d_codes, n, len, row_stride, nullptr, stream);
    DeviceSet &s = ctx->set[slot];
    dst_ctx::Shared &sh = ctx->shared[slot];
    int rc = wait_for_other_runs(ctx, stream);   // a run on another stream may still read the lists about to be replaced
    if (!rc)
        rc = ensure_lut(ctx);
    if (!rc)
        rc = shape_set(ctx, s, n, len);
    if (!rc)
        rc = alloc_ref(ctx, s);
    if (rc)
        return rc;
    // entries per block: 1.25 x the largest block of the previous shared upload of this slot (known to every rank), else
    // 96 per record to begin with
    SharedLayout lay = shared_layout(n, world, 0);
    const uint64_t want_cap = sh.last_biggest ? sh.last_biggest + sh.last_biggest / 4 + 4096 : (uint64_t)lay.rmax * 96 + 16384;
    if (want_cap * (uint64_t)world >= 0x7FFFFFF0ull)
        return pack_from_device(ctx, slot, d_codes, n, len, row_stride, nullptr, stream);
    lay = shared_layout(n, world, (uint32_t)want_cap);
    const size_t rec_begin = std::min<size_t>(n, (size_t)rank * lay.rmax), rec_end = std::min<size_t>(n, ((size_t)rank + 1) * lay.rmax);
    const size_t count = rec_end - rec_begin;
    const size_t block_bytes = (size_t)lay.words * sizeof(uint32_t);
    // buffers: the pack's counts and slots (like pack_queue), the exchange blocks, the CSR of the whole set
    if (s.rec.pre_cap < n + 1) {
        if (s.rec.pre_cold)
            HIP_TRY(ctx, hipFree(s.rec.pre_cold));
        s.rec.pre_cold = s.rec.pre_hot = nullptr;
        s.rec.pre_totals = nullptr;
        s.rec.pre_cap = 0;
        const size_t words = 5 * (n + 1) + 8 + ((2 * (n + 1)) & 1);   // (pack_queue's layout: the run-chunk counters behind the totals)
        HIP_TRY(ctx, hipMalloc((void **)&s.rec.pre_cold, words * sizeof(uint32_t)));
        s.rec.pre_cap = n + 1;
    }
    {
        const size_t cap = s.rec.pre_cap, pad = (2 * cap) & 1;
        s.rec.pre_hot = s.rec.pre_cold + cap;
        s.rec.pre_totals = reinterpret_cast<unsigned long long *>(s.rec.pre_cold + 2 * cap + pad);
    }
    const size_t n_ranges = (s.nchunks * kChunkSites + kBucketSites - 1) / kBucketSites;
    rc = ensure_bytes(ctx, (void **)&s.rec.pre_slots, &s.rec.pre_slots_cap, s.nchunks * s.npad * sizeof(uint4));
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&sh.send, &sh.send_bytes, block_bytes);
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&sh.recv, &sh.recv_bytes, block_bytes * (size_t)world);
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&sh.off_local, &sh.off_local_bytes, (lay.rmax + 2) * sizeof(uint32_t));
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&s.rec.off, &s.rec.off_cap, (n + 1) * sizeof(uint32_t));
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&s.rec.ent, &s.rec.ent_cap, ((size_t)lay.ent_cap * world + 4) * sizeof(uint32_t));
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&s.site.ent, &s.site.ent_cap, (size_t)lay.ent_cap * world * sizeof(uint32_t));
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&s.rec.range_start, &s.rec.range_cap, n_ranges * s.npad * sizeof(uint32_t));
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&ctx->scan_tmp, &ctx->scan_tmp_bytes, scan_tmp_words(n + 1) * sizeof(uint32_t));
    if (rc)
        return rc;
    // ---- this rank's share: reference from the bytes, pack + count, scan, lists straight into the block
    // (the pack's counts and first-invalid-byte cell are cleared on the sample's way: no fills of their own)
    HIP_TRY(ctx, launch_ref_sample_bytes(d_codes, row_stride, s, stream, s.rec.pre_cold, 5 * s.rec.pre_cap + ((2 * s.rec.pre_cap) & 1) + 8,
                                         ctx->d_first_bad));
    HIP_TRY(ctx, launch_hot_list(s, stream));
    PackLists pl{};
    pl.ref_planes = s.ref.planes;
    pl.hot_planes = s.ref.hot_planes;
    pl.stats = reinterpret_cast<const unsigned long long *>(s.ref.stats);
    pl.max_dev_sum = (unsigned long long)(kListsMaxDeviation * (double)len * (double)std::min<size_t>(n, kRefSamples));
    pl.cnt_cold = s.rec.pre_cold;
    pl.cnt_hot = s.rec.pre_hot;
    pl.slots = s.rec.pre_slots;
    pl.defer_planes = planes_deferred_by_pack() ? 1 : 0;   // (a partial set runs on the consensus path only: nothing ever reads them)
    rc = timer_begin(ctx, 1, stream);
    if (rc)
        return rc;
    HIP_TRY(ctx, launch_pack(d_codes, row_stride, s, ctx->d_first_bad, &pl, stream, rec_begin, rec_end));
    rc = timer_end(ctx, 1, stream);
    if (rc)
        return rc;
    uint32_t *block = static_cast<uint32_t *>(sh.send);
    HIP_TRY(ctx, launch_exclusive_scan(sh.off_local, count + 1, ctx->scan_tmp, stream, s.rec.pre_cold + rec_begin, s.rec.pre_hot + rec_begin));
    HIP_TRY(ctx, launch_shared_block(block, lay, s, rec_begin, count, sh.off_local, ctx->d_first_bad, stream));
    HIP_TRY(ctx, launch_slot_fill(s, s.ref.planes, s.ref.hot_planes, false, sh.off_local, block + lay.ent_at, nullptr, stream,
                                  rec_begin, rec_end, lay.ent_cap));
    if (with_counts)
        HIP_TRY(ctx, launch_range_counts(s, rec_begin, rec_end, block + lay.counts_at, stream, &pl));
    // ---- the exchange, and the lists of every record into the set's CSR
    rc = comm_allgather(comm, sh.send, sh.recv, block_bytes, stream);
    if (rc)
        return rc;
    HIP_TRY(ctx, launch_shared_splice(static_cast<const uint32_t *>(sh.recv), lay, s, s.rec.pre_cold, ctx->scan_tmp, with_counts != 0,
                                      ctx->d_report, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    const unsigned long long first_bad = ctx->h_report[0], total = ctx->h_report[9], over = ctx->h_report[10];
    for (int k = 0; k < 8; ++k)
        s.ref.h_stats[k] = ctx->h_report[1 + k];
    sh.last_biggest = ctx->h_report[11];
    if (first_bad != ~0ull)
        return invalid_code_error(ctx, first_bad, len);
    const double max_dev = kListsMaxDeviation * (double)len * (double)std::min<size_t>(n, kRefSamples);
    const bool diverse = (double)s.ref.h_stats[1] > (double)(unsigned long long)max_dev;   // the pack kernel's own test: no slots written
    // hot columns that carry real weight (clade-defining sites: more than half an event per pair, a quarter of what a
    // result costs to write) want the hybrid path, which needs every record's planes on every rank
    const double samples = (double)std::max<uint64_t>(s.ref.h_stats[3], 1);
    const bool hot_columns = (double)(s.ref.h_stats[2] - s.ref.h_stats[7]) / (samples * samples) > 0.5;
    if (over || diverse || hot_columns || total > 0x7FFFFFFFull / 4) {
        sh.fallbacks += 1;
        return pack_from_device(ctx, slot, d_codes, n, len, row_stride, nullptr, stream);   // every rank, by the same figures
    }
    s.loaded = true;
    s.partial = true;
    s.part_begin = rec_begin;
    s.part_end = rec_end;
    s.lean = true;
    s.planes_deferred = pl.defer_planes != 0;
    s.have_counts = with_counts != 0;
    s.ref.valid = true;
    s.rec.pre_valid = false;           // the slots cover this rank's records only
    s.rec.total = total;
    s.rec.valid = true;
    s.rec.ranges_valid = true;
    s.rec.ref_owner = &s;
    s.rec.ref_epoch = s.epoch;
    s.rec.without_hot = false;
    s.site.valid = false;
    s.aconst_family = -1;
    sh.uploads += 1;
    return publish_prep(ctx, stream);
}

extern "C" int dst_shared_stats(const dst_ctx *ctx, int slot, uint64_t *shared_uploads, uint64_t *fallbacks, uint64_t *block_entries)
{
    if (!ctx || slot < 0 || slot > 1)
        return DST_ERR_ARG;
    if (shared_uploads)
        *shared_uploads = ctx->shared[slot].uploads;
    if (fallbacks)
        *fallbacks = ctx->shared[slot].fallbacks;
    if (block_entries)
        *block_entries = ctx->shared[slot].last_biggest;
    return DST_OK;
}
