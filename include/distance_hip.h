/*
 * distance_hip.h — C ABI of libdistance_hip.so, the MI355X (gfx950) all-pairs genetic-distance
 * engine.  Drop-in for the hot path of benjamincjackson/distance: everything between
 * set_up()'s outputs and gather_write()'s input (src/lib.rs:269-474, 502-596), i.e. the pair
 * generator + worker pools that call `fn(&EncodedFastaRecord, &EncodedFastaRecord) -> FloatInt`
 * (src/lib.rs:477-488, call sites src/lib.rs:325 and :434).
 *
 * The reference has no FFI of its own; INTEGRATION.md shows the Rust `extern "C"` block a
 * maintainer would add.  Everything here is POD: plain pointers and sizes, no C++/torch types.
 *
 * Conventions
 *  - every entry point returns a dst_status (0 = ok) and never throws or aborts;
 *    dst_last_error() gives the message of the last failure on that context.
 *  - a context is bound to ONE GPU and is single-owner (not re-entrant).  One process per GPU.
 *  - the caller owns every pointer it passes; host pointers are not retained after return.
 *  - "codes" are Paradis bytes exactly as src/encoding.rs:4-41 produces them (17 valid values);
 *    any other byte is rejected by dst_upload* with DST_ERR_INVALID_CODE (the reference rejects
 *    the character earlier, in encode(), src/fastaio.rs:111-113).
 *  - canonical pair order = the reference's: square i<j row-major (src/lib.rs:511-512),
 *    rectangle i outer / j inner (src/lib.rs:560-561).  Stream mode (src/lib.rs:322-331:
 *    streamed record outer, loaded record inner) is a rectangle run with the streamed batch as
 *    the row set, because every measure is symmetric in its two records.
 */
#ifndef DISTANCE_HIP_H
#define DISTANCE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DST_ABI_VERSION 3

typedef struct dst_ctx dst_ctx;

/* -m values, src/lib.rs:104-109; dispatch table of get_distance_function, src/lib.rs:477-488 */
typedef enum {
    DST_N = 0,      /* snp_consensus, src/measures.rs:28-53 (same integers as n_high, computed densely) */
    DST_N_HIGH = 1, /* snp,           src/measures.rs:14-23  */
    DST_RAW = 2,    /* raw,           src/measures.rs:56-69  */
    DST_JC69 = 3,   /* jc69,          src/measures.rs:72-77  */
    DST_K80 = 4,    /* k80,           src/measures.rs:80-113 */
    DST_TN93 = 5    /* tn93,          src/measures.rs:116-193 */
} dst_measure;

typedef enum {
    DST_OK = 0,
    DST_ERR_ARG = 1,          /* bad argument / null pointer / range */
    DST_ERR_HIP = 2,          /* a HIP runtime call failed (message has hipGetErrorString) */
    DST_ERR_INVALID_CODE = 3, /* a byte that src/encoding.rs never produces */
    DST_ERR_STATE = 4,        /* set not uploaded, widths differ (src/fastaio.rs:93-95), ... */
    DST_ERR_NOMEM = 5,
    DST_ERR_CAPACITY = 6      /* output buffer too small */
} dst_status;

/* What a run writes per pair, in canonical order:
 *  DST_OUT_DISTANCE: 8 bytes — the payload of FloatInt (src/measures.rs:5-9): int64 for n/n_high,
 *                    f64 for raw/jc69/k80/tn93 finalised ON DEVICE in the pair kernel's epilogue: raw is the
 *                    reference's bits (one correctly rounded division); jc69 / k80 / tn93 are within 1e-12 of the
 *                    reference (series logarithms and reciprocal multiplies where every logarithm's argument lies
 *                    within 2^-5 of 1, the reference's operation order with a table logarithm elsewhere), not
 *                    bit-identical: for the reference's bits take DST_OUT_TALLY + dst_finalize, for its TEXT dst_text_*.
 *  DST_OUT_TALLY:    dst_tally_width(measure) x uint32 site tallies, bit-exact integers:
 *                      n, n_high : {d}
 *                      raw, jc69 : {n, d}                        (src/measures.rs:57-66)
 *                      k80       : {count_L, ts, tv}             (src/measures.rs:81-107)
 *                      tn93      : {count_L, count_d, count_P1, count_P2} (src/measures.rs:150-175)
 *                    finalise with dst_finalize() on the host for bit-identical TSV text.
 *  DST_OUT_TALLY16:  the same tallies as uint16 (2 x width bytes per pair) for alignments shorter
 *                    than 65,536 sites: the compact form sent between GPUs; dst_finalize_device()
 *                    turns it into distances on the receiving GPU. */
typedef enum { DST_OUT_DISTANCE = 0, DST_OUT_TALLY = 1, DST_OUT_TALLY16 = 2 } dst_output;
/* OR-ed into dst_finalize_device's tally_kind: finalise in the reference's operation order with the device's table
 * logarithm (within a few ulp of the host's libm: what dst_text_* prints from) instead of the epilogue's arithmetic */
#define DST_FIN_CLOSE 0x100

/* ---- library ------------------------------------------------------------------------- */
int dst_abi_version(void);
int dst_device_count(int *count);
/* name -> dst_measure, or -1 (src/lib.rs:477-488 panics on unknown names; clap rejects first) */
int dst_measure_from_name(const char *name);
int dst_tally_width(int measure);
const char *dst_status_string(int status);
/* The measurement macros (DST_DBG_*: alternative store patterns, wave splits ... of tools/variants.sh) this library was
 * compiled with, space separated; "" for a production build.  bench.py refuses to time anything else. */
const char *dst_build_flags(void);

/* ---- context -------------------------------------------------------------------------- */
int dst_create(int device, dst_ctx **ctx);
int dst_destroy(dst_ctx *ctx);
const char *dst_last_error(const dst_ctx *ctx); /* ctx may be NULL: last dst_create failure */
/* kernel tile variant: 0 = default for the measure; see DESIGN.md "tile variants" */
int dst_set_variant(dst_ctx *ctx, int variant);
int dst_variant_count(int measure);
/* split-L factor: launches with few tiles and a long alignment (small sets, streamed batches) cut
 * the sweep over L into `ksplit` parts whose partial integer tallies are combined with atomics
 * (exact).  0 = automatic (default: launches with fewer than 1,024 tiles), 1 = never split, k > 1 = force. */
int dst_set_ksplit(dst_ctx *ctx, int ksplit);

/* DST_PATH_AUTO leaves small jobs to the dense kernels without sampling them: an upload with fewer than
 * `site_comparisons` (n^2/2 x len) ahead of it skips the consensus path's preparation, and a run below it (with no
 * reference sequence yet) goes dense.  Default 2e10 (what the dense kernels finish in ~0.1 ms); 0 makes every upload
 * prepare the lists (tests drive the small shapes of the parity suite through the fused preparation this way). */
int dst_set_prep_threshold(dst_ctx *ctx, double site_comparisons);
/* Which kernels a run uses.  Both give the same integers (tallies bit-exact, the same finalisation):
 *  DST_PATH_DENSE:     bit-plane tile kernels, work ~ pairs x L whatever the data (src/measures.rs:14-23,
 *                      56-66, 85-107, 156-175 evaluated at every site).
 *  DST_PATH_CONSENSUS: the idea of the reference's `-m n` (snp_consensus, src/measures.rs:28-53, over the
 *                      difference lists of get_differences(), src/fastaio.rs:67-75, against consensus(),
 *                      src/fastaio.rs:289-336) carried to every measure: tallies from each record's
 *                      differences to a per-site plurality sequence; work ~ pairs + the differences two
 *                      records share.  Low-diversity alignments (SARS-CoV-2-like) run output-bound.
 *                      Shapes its lists cannot index (zero-width alignments, 2^28 or more records or sites,
 *                      more than 2^31 differences in a set) run dense even when this is selected.
 *  DST_PATH_HYBRID:    consensus path for the "cold" columns, dense kernels for the hot ones (columns where more than
 *                      5 % of a sample of the records deviate from the plurality: clade-defining mutations), whose
 *                      tallies the consensus kernel adds in.  Alignments with phylogenetic structure stay fast.
 *                      Without hot columns it is the consensus path; with mostly hot columns the dense one.
 *  DST_PATH_AUTO:      (default) per launch, whichever a sampled estimate of the alignment's diversity
 *                      says is fastest. */
typedef enum { DST_PATH_AUTO = 0, DST_PATH_DENSE = 1, DST_PATH_CONSENSUS = 2, DST_PATH_HYBRID = 3 } dst_path;
int dst_set_path(dst_ctx *ctx, int path);
/* Records with long runs of N (failed amplicons, partial genomes; N adds nothing to any tally: src/measures.rs:17, 59-66,
 * 89-107, 160-175) in the set of `slot`: how many the consensus path currently treats as "run records" — their chunks
 * of 128 N sites are left out of the difference lists and every pair with such a record is corrected exactly
 * (DESIGN.md 3b'') — and how many list entries that removed.  0 when the set has none, too many (more than a third of the
 * records), or its lists were not built by the upload's fused preparation.  Diagnostic; the results do not depend on it. */
int dst_run_records(const dst_ctx *ctx, int slot, uint64_t *run_records, uint64_t *entries_removed);
/* *stored = 1 when the bit-planes of every (record, 128-site chunk) of the set are in HBM; 0 while the upload has deferred them: a
 * set prepared for the consensus path keeps, per chunk, its differences from the set's reference sequence (what that path
 * reads), and the planes of a chunk are written only if it does not fit that form — the rest is written, from the
 * reference and the differences, the first time something reads planes (a dense or hybrid run, dst_consensus,
 * dst_differences, a run against another set).  Diagnostic; the results do not depend on it. */
int dst_planes_stored(const dst_ctx *ctx, int slot, int *stored);
/* DST_PATH_DENSE, DST_PATH_CONSENSUS or DST_PATH_HYBRID: what the most recent run on this context used */
int dst_last_path(const dst_ctx *ctx);

/* ---- input: replaces Setup.loaded_fastas[slot] (src/lib.rs:133-144) --------------------- */
/* codes: row-major n x len Paradis bytes, rows row_stride bytes apart (>= len).
 * base_counts: n x 4 {A, T, G, C} per-record counts for tn93 (src/fastaio.rs:53-66, or the
 * streamed variant src/fastaio.rs:136-142), or NULL to have them counted on device by code.
 * slot is 0 or 1.  Both slots must have the same len (src/fastaio.rs:206-208). */
int dst_upload(dst_ctx *ctx, int slot, const uint8_t *codes, size_t n, size_t len,
               size_t row_stride, const uint32_t *base_counts);
/* same with codes / base_counts already in this GPU's memory; `stream` is a hipStream_t (NULL =
 * the context's own stream).  Asynchronous on that stream except for the validity check. */
int dst_upload_device(dst_ctx *ctx, int slot, const void *d_codes, size_t n, size_t len,
                      size_t row_stride, const uint32_t *d_base_counts, void *stream);
int dst_set_info(const dst_ctx *ctx, int slot, size_t *n, size_t *len);
/* per-record {A,T,G,C} counts the device holds for `slot` (n x 4), copied to host */
int dst_get_base_counts(dst_ctx *ctx, int slot, uint32_t *counts);

/* ---- per-alignment precompute of `-m n` (src/lib.rs:223-231) ------------------------------ */
/* consensus(), src/fastaio.rs:289-336, computed on the device: per site the plurality of A, G, C, T over
 * every record of slot 0 (and of slot 1 too when both_slots != 0 and it is loaded — the reference walks every
 * loaded file), every other code counted as A, ties to the first of A, G, C, T.  cons receives len codes
 * (136 / 72 / 40 / 24). */
int dst_consensus(dst_ctx *ctx, int both_slots, uint8_t *cons, size_t cap);
/* get_differences(), src/fastaio.rs:67-75, for every record of `slot` against `other` (len codes, host
 * memory; normally the consensus): ascending sites with seq[i] < 240 && seq[i] != other[i].  CSR output:
 * offsets has n + 1 entries, sites holds offsets[n] entries.  Pass sites == NULL to get offsets / *total only;
 * DST_ERR_CAPACITY when cap_sites < *total. */
int dst_differences(dst_ctx *ctx, int slot, const uint8_t *other, size_t len, uint64_t *offsets, uint32_t *sites,
                    size_t cap_sites, uint64_t *total);

/* ---- canonical order helpers (src/lib.rs:502-596) --------------------------------------- */
uint64_t dst_square_pairs(uint64_t n);                   /* n(n-1)/2 */
uint64_t dst_square_row_start(uint64_t n, uint64_t i);   /* index of pair (i, i+1) */
/* cut rows [0, n) into `parts` contiguous ranges of near-equal PAIR count (square) — the
 * multi-GPU partition.  bounds has parts+1 entries, bounds[0]=0, bounds[parts]=n. */
int dst_partition_square(uint64_t n, int parts, uint64_t *bounds);
int dst_partition_rect(uint64_t n_rows, int parts, uint64_t *bounds);

/* ---- run: replaces generate_pairs_* + the worker pools (src/lib.rs:367-474, 269-365) ----- */
/* All pairs (i, j), row_begin <= i < row_end, i < j < n of slot 0, canonical order, written to
 * d_out (device memory) starting with pair (row_begin, row_begin+1).  Asynchronous on `stream`
 * (hipStream_t; NULL = context stream, then the call synchronises before returning). */
int dst_run_square(dst_ctx *ctx, int measure, uint64_t row_begin, uint64_t row_end, int out_kind,
                   void *d_out, size_t out_capacity_bytes, void *stream);
/* All pairs (i, j), i in rows [row_begin,row_end) of row_slot, j over every record of col_slot;
 * out[(i-row_begin) * n_col + j].  Two loaded files: row_slot=0, col_slot=1 (src/lib.rs:432-433).
 * Stream mode: row_slot = the streamed batch, col_slot = the loaded set (src/lib.rs:322-331). */
int dst_run_rect(dst_ctx *ctx, int measure, int row_slot, int col_slot, uint64_t row_begin,
                 uint64_t row_end, int out_kind, void *d_out, size_t out_capacity_bytes,
                 void *stream);
/* Tallies (DST_OUT_TALLY or DST_OUT_TALLY16 layout, canonical order of rows [row_begin,row_end))
 * that are already in this GPU's memory -> the DST_OUT_DISTANCE payload in d_out, with the same
 * device arithmetic as a direct DST_OUT_DISTANCE run (bitwise the same values; tally_kind | DST_FIN_CLOSE: the text
 * path's arithmetic instead).  The two sets must
 * be uploaded on this context (tn93 reads their base counts).  Multi-GPU: rank 0 finalises the
 * compact tallies it gathered from the other ranks. */
int dst_finalize_device(dst_ctx *ctx, int measure, int square, int row_slot, int col_slot,
                        uint64_t row_begin, uint64_t row_end, int tally_kind, const void *d_tallies,
                        void *d_out, size_t out_capacity_bytes, void *stream);
/* host-buffer forms: run + copy back (h_out is ordinary or pinned host memory) */
int dst_run_square_host(dst_ctx *ctx, int measure, uint64_t row_begin, uint64_t row_end,
                        int out_kind, void *h_out, size_t out_capacity_bytes);
int dst_run_rect_host(dst_ctx *ctx, int measure, int row_slot, int col_slot, uint64_t row_begin,
                      uint64_t row_end, int out_kind, void *h_out, size_t out_capacity_bytes);
/* ---- stream mode: replaces stream()'s worker pool (src/lib.rs:269-365) ----------------------- */
/* Batches of streamed records (rows) against the loaded set of slot 0 (columns), results in the reference's
 * streamed-major order [streamed record][loaded record] (src/lib.rs:322-331), through a ring of `depth` slots
 * (2..16) on three HIP streams: while batch k is compared, batch k+1 crosses PCIe from page-locked memory and
 * batch k-1's results travel back.  out_kind: DST_OUT_DISTANCE or DST_OUT_TALLY.  max_records: records per batch
 * (stream_fasta()'s batchsize, src/fastaio.rs:215-286).  The width is slot 0's; slot 0 must stay loaded and
 * unchanged while the stream is open.  One owner thread, like the context. */
typedef struct dst_stream dst_stream;
int dst_stream_open(dst_ctx *ctx, int measure, int out_kind, size_t max_records, int depth, dst_stream **stream);
/* The same with a choice of what crosses the host link (which bounds a streamed job: 64 records of 5 Mbp are 320 MB):
 *  DST_WIRE_CODES    Paradis bytes, one per site (dst_stream_open);
 *  DST_WIRE_NIBBLES  the HIGH NIBBLE of each code — all that any measure reads — two sites per byte, site 2k in the low
 *                    four bits of byte k, site 2k + 1 in the high four; a row's unused trailing nibble is ignored.  The
 *                    host writes encoding_array()[c] >> 4 (src/encoding.rs:4-41) instead of the code; 0 is not a code
 *                    (DST_ERR_INVALID_CODE at collect).  Half the bytes, same results; base counts by code
 *                    (use_base_counts == 0) are counted from the nibbles on the device like from the codes. */
typedef enum { DST_WIRE_CODES = 0, DST_WIRE_NIBBLES = 1 } dst_wire;
int dst_stream_open_wire(dst_ctx *ctx, int measure, int out_kind, size_t max_records, int depth, int wire, dst_stream **stream);
/* The page-locked input buffer of the next batch: rows *pitch bytes apart (>= width — half of it, rounded up, for
 * DST_WIRE_NIBBLES — and a multiple of 128), room for max_records; encode straight into it.  *base_counts (may be NULL): max_records x 4 {A,T,G,C} for tn93
 * (encode_count_bases(), src/fastaio.rs:120-145).  DST_ERR_STATE when every slot is in flight. */
int dst_stream_acquire(dst_stream *stream, uint8_t **codes, size_t *pitch, uint32_t **base_counts);
/* Queue the acquired buffer holding n_records records: H2D, pack, compare, D2H.  Returns without waiting.
 * use_base_counts != 0: tn93 uses the caller's counts (streamed records: upper-case letters only,
 * src/fastaio.rs:136-142), else they are counted on the device by code. */
int dst_stream_submit(dst_stream *stream, size_t n_records, int use_base_counts);
/* Wait for the OLDEST submitted batch and hand out its results (page-locked, library-owned, valid until the next
 * dst_stream_submit): n_records x (records of slot 0) x the per-pair payload of out_kind.
 * DST_ERR_INVALID_CODE when the batch held a byte src/encoding.rs never produces. */
int dst_stream_collect(dst_stream *stream, size_t *n_records, const void **results);
int dst_stream_in_flight(const dst_stream *stream);   /* submitted, not yet collected */
int dst_stream_close(dst_stream *stream);

/* ---- multi-GPU: the gather of the result slabs (one process per GPU, RCCL over xGMI) --------- */
/* The pair space shards by contiguous canonical ranges (dst_partition_square / dst_partition_rect): every rank
 * uploads the whole set, runs its own row range, and the only exchange is this gather into ONE rank's buffer
 * — what gather_write consumes (src/lib.rs:612-644).  Grouped ncclSend / ncclRecv straight from each rank's slab
 * into its place in the root's buffer (peer -> root on all xGMI links at once; no ring, no staging).  RCCL is
 * loaded on first use (librccl.so.1); single-GPU hosts never touch it.
 * Bootstrap: rank 0 calls dst_comm_unique_id and hands the DST_COMM_ID_BYTES bytes to the other ranks by whatever
 * its launcher offers (environment, file, MPI ...); then every rank calls dst_comm_create (collective). */
#define DST_COMM_ID_BYTES 128
typedef struct dst_comm dst_comm;
int dst_comm_unique_id(uint8_t *id, size_t cap);
int dst_comm_create(dst_ctx *ctx, const uint8_t *id, int rank, int world, dst_comm **comm);
/* A communicator over the caller's own transport (MPI, a test harness): `allgather` must take `bytes_per_rank` bytes at
 * d_send on every rank and leave rank r's at d_recv + r * bytes_per_rank on every rank (device memory), ordered after
 * the work already queued on `stream` (hipStream_t) and visible to work queued on it afterwards; 0 = ok.  Serves
 * dst_upload_shared; dst_gather_slabs needs an RCCL communicator. */
typedef int (*dst_allgather_fn)(void *user, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream);
int dst_comm_create_custom(dst_ctx *ctx, int rank, int world, dst_allgather_fn allgather, void *user, dst_comm **comm);
int dst_comm_destroy(dst_comm *comm);
int dst_comm_info(const dst_comm *comm, int *rank, int *world);
/* Collective.  byte_offsets / byte_sizes have `world` entries and are the same on every rank: rank r contributes
 * byte_sizes[r] bytes from its d_local, which land at d_full + byte_offsets[r] on `root` (d_full may be NULL on
 * the other ranks; the root's own slab is copied only if it is not already in place).  Any payload: f64 / int64
 * results, or DST_OUT_TALLY16 tallies that the root then finalises with dst_finalize_device.  A rank's range may
 * be sent in several calls (sub-slabs): the transfer of sub-slab k runs on `stream` behind what is already queued
 * there, so the next dst_run_square on another stream overlaps it.  Asynchronous on `stream` (hipStream_t;
 * NULL = the context's stream, then the call waits). */
int dst_gather_slabs(dst_comm *comm, const void *d_local, void *d_full, const uint64_t *byte_offsets,
                     const uint64_t *byte_sizes, int root, void *stream);

/* ---- multi-GPU: the preparation of a loaded set shared out over the ranks ------------------------------------ */
/* Collective form of dst_upload_device for slot 0 against itself (one file, the square job): instead of every rank
 * packing and indexing the whole set before it computes its row range — the reference's workers all read the ONE
 * prepared copy of loaded_fastas, src/lib.rs:413-458, 219-242 — rank k packs and lists records
 * [begin, end) = dst_shared_range(n, k, world) only, one all-gather brings every rank's difference lists (and base
 * counts, with_counts != 0: needed by tn93) to every rank, and the site tables and per-record constants are built from the
 * lists locally.  d_codes is the WHOLE n x len matrix in this GPU's memory (the call reads the rank's own records and the
 * 512 records every rank samples the reference sequence from).  Afterwards dst_run_square / dst_text_square work for any
 * row range on the consensus path; the dense and hybrid kernels, dst_consensus and dst_differences need every record's
 * planes and refuse such a set (DST_ERR_STATE).  When the lists cannot serve (a context forced dense, a set too
 * diverse, hot columns, lists larger than the exchange blocks — every rank reads that from the same figures) the call
 * IS dst_upload_device on every rank.  Every rank must pass the same n, len, with_counts, the same data and hold the
 * same dst_set_path / dst_set_prep_threshold settings.  Synchronous like dst_upload_device (validity check). */
int dst_upload_shared(dst_comm *comm, int slot, const void *d_codes, size_t n, size_t len, size_t row_stride,
                      int with_counts, void *stream);
/* the records rank `rank` of `world` prepares (pure host code) */
int dst_shared_range(uint64_t n, int rank, int world, uint64_t *begin, uint64_t *end);
/* The exchange block of one rank for blocks of `entries` list entries, in 32-bit words (pure host code; documentation
 * and tests — dst_upload_shared sizes its blocks itself): layout = {records per rank, offset of the list lengths, of the
 * base counts (4 per record), of the entries, entry capacity (rounded up to 4), words per block}.  Words 0..3 of a block:
 * entries in it, 1 if they did not fit, the first invalid byte's index (64 bits, ~0: none). */
int dst_shared_block_layout(uint64_t n, int world, uint32_t entries, uint32_t layout[6]);
/* how dst_upload_shared went on this context so far: uploads that were shared, uploads that fell back to the replicated
 * form, entries of the largest exchange block of the last one (any pointer may be NULL) */
int dst_shared_stats(const dst_ctx *ctx, int slot, uint64_t *shared_uploads, uint64_t *fallbacks, uint64_t *block_entries);

/* In-order sink (the shape of gather_write's input, src/lib.rs:612-644): the run is cut into row
 * slabs of at most max_pairs pairs (>= one row each) and `sink` is called once per slab, strictly
 * in canonical order, on the calling thread, with the slab's results in library-owned pinned host
 * memory that is valid only during the call.  While the sink works on slab k the GPU already computes
 * slab k+1 and its copy back is in flight.  A non-zero return from the sink stops the run
 * (DST_ERR_STATE, message "stopped by sink").  square != 0: slot 0 against itself (row_slot/col_slot
 * ignored). */
typedef int (*dst_slab_sink)(void *user, uint64_t first_pair, uint64_t n_pairs, uint64_t row_begin,
                             uint64_t row_end, const void *data);
int dst_run_slabs(dst_ctx *ctx, int measure, int square, int row_slot, int col_slot, int out_kind,
                  uint64_t max_pairs, dst_slab_sink sink, void *user);
/* Page-locked host memory for the *_host forms' output buffers (copy-back by DMA at link speed instead
 * of through a pageable bounce buffer).  Free with dst_host_free. */
int dst_host_alloc(size_t bytes, void **ptr);
int dst_host_free(void *ptr);
/* bytes a run writes */
size_t dst_out_bytes(int measure, int out_kind, uint64_t n_pairs);
/* milliseconds of the pair kernel of the most recent run and of the pack kernel of the most recent
 * upload on this context, from HIP events recorded on the launch stream (bench.py's roofline leg);
 * *finalize_ms is always 0: finalisation is fused into the pair kernel's epilogue */
int dst_last_kernel_ms(dst_ctx *ctx, float *pair_ms, float *finalize_ms, float *pack_ms);

/* The same over many launches without waiting for the device after each: means over the pair-kernel and pack-kernel
 * launches of this context since the last call with reset != 0 (at most the 64 most recent of each; waits for the last
 * one).  bench.py times its steps with one call at the end of the timed region. */
int dst_kernel_ms_mean(dst_ctx *ctx, int reset, float *pair_ms, int *pair_launches, float *pack_ms, int *pack_launches);

/* Diagnostic: the tile schedule one pair-kernel launch would use for rows [row_begin,row_end)
 * against n_cols records — (i0, j0) per workgroup in launch order, idle fillers as i0 = 2^32-1.
 * tile_rows/tile_cols receive the tile shape of (measure, variant).  ij may be NULL to query
 * *count only.  Pure host code (no GPU needed). */
int dst_plan_tiles(int square, uint64_t row_begin, uint64_t row_end, uint64_t n_cols, int measure,
                   int variant, uint32_t *ij, size_t cap_tiles, size_t *count, int *tile_rows,
                   int *tile_cols);

/* What ONE site contributes to each tally of `measure` for the code pair (q, t): the bodies of the site loops
 * of src/measures.rs:14-23, 56-66, 85-107, 156-175 (out has dst_tally_width(measure) entries, each 0 or 1).
 * Pure host code; the consensus path's tables are built from it. */
int dst_site_tallies(int measure, uint8_t q, uint8_t t, int *out);

/* ---- host finalisation in the reference's f64 operation order --------------------------- */
/* tallies: dst_tally_width(measure) uint32 per pair.  q_counts/t_counts: {A,T,G,C} of record_1 /
 * record_2 (tn93 only, else may be NULL).  Writes the FloatInt payload: *as_int for n/n_high,
 * *as_float otherwise (glibc log/sqrt, -ffp-contract=off => bit-identical to src/measures.rs). */
int dst_finalize(int measure, const uint32_t *tallies, const uint32_t *q_counts,
                 const uint32_t *t_counts, double *as_float, int64_t *as_int);
/* One TSV field as gather_write prints it (src/lib.rs:626-633): `{}` / `{:.12}` incl. Rust's
 * "NaN", "inf", "-inf", "-0.000000000000".  snprintf semantics: returns the length of the full text
 * (no NUL counted); if that is >= cap the text was truncated to cap-1 characters. */
int dst_format_distance(int measure, double as_float, int64_t as_int, char *buf, size_t cap);

/* ---- TSV text on the device ---------------------------------------------------------------- */
/* gather_write()'s output (src/lib.rs:612-644) produced by the GPU: "id1\tid2\tvalue\n" per pair of rows
 * [row_begin, row_end) in canonical order, `{}` / `{:.12}` exactly as dst_format_distance prints one value.
 * dst_set_ids gives the record ids of the packed set in `slot` (record r: chars[offsets[r] .. offsets[r+1])); they
 * stay valid across uploads of the same record count.  The text goes to `out` (host memory; page-locked memory
 * from dst_host_alloc copies at link speed), *len receives its length.  At most 2^31 pairs, 65,535 rows and 4 GB of
 * text per call.  DST_ERR_CAPACITY: `capacity` is too small; DST_ERR_STATE: a value has no short text (|v| >= 1.8e7,
 * which no distance reaches): format those rows on the host.  dst_text_rect: swap_ids != 0 prints the column
 * record's id first (the value is the one dst_run_rect gives).
 *
 * The text is byte for byte what the reference prints.  n / n_high / raw: the device's values are the reference's bits.
 * jc69 / k80 / tn93 (f64::ln = the host's libm, src/measures.rs:76, 109-112, 187): the device finalises the pair's
 * integer tallies with its own logarithm (within a few ulp of libm's) and notes every value that lies within 2^-47 |v| of
 * a rounding boundary of the 12th decimal; those pairs (~1e-4 of a low-diversity alignment's lines) are re-finalised on
 * the host by dst_finalize and their digits overwritten before the call returns — any value within the guard of a line
 * that was NOT noted prints the same text.  DST_ERR_STATE also when more than 1/16 of a slab's values are near ties
 * (distances far above 1): format that slab on the host. */
int dst_set_ids(dst_ctx *ctx, int slot, const char *chars, const uint64_t *offsets, uint64_t n);
/* running totals over this context's dst_text_* calls: values noted as near ties, and how many of those the host's
 * finalisation printed differently from the device's (either pointer may be NULL) */
int dst_text_stats(const dst_ctx *ctx, uint64_t *near_ties, uint64_t *rewritten);
int dst_text_square(dst_ctx *ctx, int measure, uint64_t row_begin, uint64_t row_end, char *out, size_t capacity,
                    size_t *len);
int dst_text_rect(dst_ctx *ctx, int measure, int row_slot, int col_slot, uint64_t row_begin, uint64_t row_end,
                  int swap_ids, char *out, size_t capacity, size_t *len);

#ifdef __cplusplus
}
#endif
#endif /* DISTANCE_HIP_H */
