// dst_internal.h — shared between the HIP kernels (dst_kernels.hip), the C-ABI (dst_api.cpp)
// and the host-only logic (dst_host.cpp).  Not installed; the public surface is
// include/distance_hip.h.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/distance_hip.h"

namespace dst {

// ---- device-resident form of one loaded set -------------------------------------------------
// The Paradis byte matrix (src/encoding.rs) is re-laid on upload into 8 bit-planes, one bit per
// site, 128 sites (one uint4) per (plane, chunk, record):
//     planes[(plane * nchunks + chunk) * npad + record]          (uint4 = sites 128c .. 128c+127)
// so that (a) a wave whose lanes own consecutive records reads 1 KiB contiguous per plane-chunk
// and (b) the 16 bytes of one record are a wave-uniform scalar load.  Only the high nibble and
// bit 3 of a code matter to any measure, so the planes carry all of it:
enum Plane : int {
    PL_A = 0,   // code bit 7: A is a member of the site's base set
    PL_G = 1,   // code bit 6
    PL_C = 2,   // code bit 5
    PL_T = 3,   // code bit 4
    PL_K = 4,   // code bit 3: "known base" (exactly one of A,G,C,T)
    PL_X1 = 5,  // class bit: 1 for {C,T,Y} (pyrimidine class), 0 for {A,G,R}; 0 elsewhere
    PL_X0 = 6,  // within-class bit of a known base: 1 for G and T, 0 for A and C; 0 elsewhere
    PL_CL = 7,  // code is in a k80 class: {A,G,R} or {C,T,Y}
    PL_COUNT = 8
};

constexpr uint32_t kChunkSites = 128;  // sites per uint4
constexpr uint32_t kPadRecords = 512;  // npad granularity (>= every tile's BN)
constexpr uint32_t kBlockThreads = 256;

// ---- consensus-delta path (dst_consensus.hip) -------------------------------------------------
// Every tally of every measure is a sum over sites of a per-site function f_k(q[s], t[s]) of the two
// codes' high nibbles (src/measures.rs:14-23, 56-66, 85-107, 156-175).  Against a reference sequence c
// (a per-site plurality code: the idea of consensus(), src/fastaio.rs:289-336, and of snp_consensus(),
// src/measures.rs:28-53, carried to every measure):
//     T_k(q,t) = F_k + A_k(q) + A_k(t) + sum over sites where BOTH q and t differ from c of h_k
//     F_k    = sum_s f_k(c,c)              A_k(x) = sum over x's difference sites of f_k(x,c) - f_k(c,c)
//     h_k    = f_k(q,t) - f_k(q,c) - f_k(c,t) + f_k(c,c)
// exact integers whatever c is; the work is proportional to the differences from c, not to L.
constexpr uint32_t kPanelCols = 2048;      // column records per site bucket = width of the LDS accumulators
constexpr uint32_t kBucketSites = 1024;    // sites per block of site_bucket_kernel: the lists' range_start marks are this far apart
constexpr int kAccRows = 2;                // rows whose accumulators a workgroup holds at once
constexpr uint32_t kTileRowsMax = 32;      // rows of one consensus-path tile (a multiple of kAccRows)
constexpr uint32_t kEntryShift = 28;       // list entries: nibble in the top four bits
constexpr uint32_t kEntryMask = (1u << kEntryShift) - 1;   // bucket entries: column record | nibble << 28
constexpr uint32_t kSiteBits = 25;         // record-list entries: site | reference class << 25 | nibble << 28
constexpr uint32_t kSiteMask = (1u << kSiteBits) - 1;
constexpr uint32_t kInlineEvents = 15;     // bucket entries held inside the 32-byte lookup-table entry
constexpr uint32_t kInlineOverflowing = 13;  // ... of a larger bucket: its last word is where the rest are
constexpr uint32_t kSlotEntries = 7;        // differences a (record, chunk) slot of the pack holds
constexpr uint32_t kRefSamples = 512;      // records sampled for the reference sequence
constexpr int kRefClasses = 5;             // reference nibbles: A(8) G(4) C(2) T(1) N-class(15)
constexpr int kMaxWords = 4;               // packed accumulator words per pair
constexpr uint32_t kHotPermille = 50;      // hybrid path: a site is "hot" when more than 5 % of the sampled records deviate
                                           // (p^2 x 1.85 ps per event against 1 / 1.95e14 s per dense site and pair; 33 while an event cost 4.5 ps)

struct ConsensusRef {             // the reference sequence, sampled from the set that owns it
    uint4 *planes = nullptr;      // [4][nchunks] A,G,C,T planes of it (chunk-packed like the records'); N past len
    uint4 *hot_planes = nullptr;  // [nchunks] bit = 1: a hot site (kHotPermille), handed to the dense kernels by the hybrid path
    uint32_t *hot_sites = nullptr;  // [n_hot] the hot sites, ascending
    uint32_t *partials = nullptr;   // [nchunks][2][8] every chunk's two shares of `stats` (summed by hot_list_kernel)
    // device: {known sites, sum of deviants, sum of deviants^2, sample size, hot sites, known hot sites,
    //          sum of deviants over the cold sites, sum of deviants^2 over the cold sites}
    uint64_t *stats = nullptr;
    uint64_t h_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t nchunks = 0;
    bool valid = false;
};

struct RecordIndex {              // per record: the sites where it differs from the reference, ascending
    uint32_t *off = nullptr;      // [n + 1]
    uint32_t *ent = nullptr;      // site | class of the reference there << 25 | the record's nibble << 28
    // list lengths counted by the pack itself against the set's own reference (cold sites / hot sites apart), with
    // their sums {cold, hot} — valid while pre_epoch == the set's epoch
    uint4 *pre_slots = nullptr;   // [nchunks][npad] the entries themselves, by (record, chunk) (PackLists::slots)
    size_t pre_slots_cap = 0;
    uint32_t *pre_cold = nullptr, *pre_hot = nullptr;
    unsigned long long *pre_totals = nullptr;
    size_t pre_cap = 0;
    uint64_t pre_epoch = 0, pre_total_cold = 0, pre_total_hot = 0;
    bool pre_valid = false;
    size_t off_cap = 0, ent_cap = 0, range_cap = 0;
    // column sets: [ceil(nchunks / 8)][npad] where the record's list crosses every multiple of kBucketSites sites
    uint32_t *range_start = nullptr;
    uint64_t total = 0;
    const void *ref_owner = nullptr;  // the DeviceSet whose reference these lists are relative to
    uint64_t ref_epoch = 0;
    bool without_hot = false;         // hybrid path: the hot sites are left out
    bool valid = false;
    bool ranges_valid = false;        // range_start holds the marks of THESE lists (a column set's site buckets can be built from them)
};

// ---- records with long runs of N (failed amplicons, partial genomes) ----------------------------------------------
// Every site of a run is a difference from the reference sequence, and two such records share an event at every site
// where their runs overlap: 5 % of 50,000 records half N cost the consensus path 59 ms instead of 2.  N contributes
// NOTHING to any tally, whatever the other record holds, so a whole 128-site chunk of N in record t ("run chunk": the
// reference has at least one known site there) can leave t's list, if every pair (q, t) is corrected for what the
// identity then counts wrongly on those sites:
//     T(q,t) = F + A'(q) + A'(t) + H(q,t) + X(q,t)          M_x = the run chunks of x (records with fewer than kRunMin keep
//     A'(x)  = A(x: stripped list) - F(M_x)                        their entries: M_x is empty)
//     X(q,t) = F(M_q n M_t) - A_q(M_t) - A_t(M_q)             A_q(M) = sum of a(e) over q's entries in the chunks of M
// exact integers again (dst_consensus.hip: corr_kernel computes X's terms per (run record, record) once per set, the
// pair kernel's event waves add them to the accumulators).  Records with runs: "run records" (hot records in DESIGN.md).
constexpr uint32_t kRunMin = 4;            // run chunks (512 sites of N) that make a record a run record
struct RunIndex {
    uint32_t *cnt_run = nullptr;     // [n] run chunks of every record          } counted by the pack, one block with
    uint32_t *run_cold = nullptr;    // [n] their entries at cold sites         } the arrays below: n_alloc records
    uint32_t *run_hot = nullptr;     // [n] ... at hot sites
    uint32_t *index = nullptr;       // [n] run record number of a record, or 0xFFFFFFFF
    uint32_t *ids = nullptr;         // [n] the run records, ascending
    uint32_t *mask = nullptr;        // [n_run][mask_words] bit c: chunk c of the run record is a run chunk (from the slots' flags),
                                     // then the same transposed, [mask_words][n_run]
    uint32_t *known = nullptr;       // [nchunks] known reference sites of every chunk (cold sites only: without_hot lists)
    uint32_t *panel_first = nullptr; // [n_panels + 1] first run record of every column panel
    uint32_t *state = nullptr;       // device: [0] run records, [1] 1: stripping is on for this upload
    uint32_t *aent = nullptr;        // [kMaxWords][entries] a-words of every list entry (aconst_kernel), what the tables sum
    uint8_t *s7 = nullptr;           // [words][5][tiles of 32 records][mask_words][1 KB] every record's per-chunk sums of them, in 7-bit pieces
    uint32_t *corr = nullptr;        // [words][n_run][n]   X's terms by (run record, record) ...
    uint32_t *corr_t = nullptr;      // [words][n][n_run]   ... and transposed
    size_t n_alloc = 0, mask_words = 0, mask_cap = 0, known_cap = 0, panel_cap = 0, aent_cap = 0, corr_cap = 0, corr_t_cap = 0, s7_cap = 0;   // (capacities in bytes)
    uint64_t removed = 0;            // entries the run records' lists lost (host copy: rescales the sample's statistics)
    uint32_t n_run = 0;              // host copy (0: no run records, or stripping off)
    bool active = false;             // this upload's lists are stripped of the run records' run chunks
    int corr_family = -1;            // what corr / corr_t hold (like aconst)
    bool corr_wide = false, corr_without_hot = false;
    uint64_t corr_epoch = 0;
};

struct SiteIndex {                // the same entries of a column set by (site, panel of kPanelCols records)
    // [n_sites * n_panels] 32 bytes per bucket = 16 halfwords: [0] entries in the bucket, then up to kInlineEvents
    // entries as record-in-panel | nibble << 11.  The pair kernel reads THIS: one request per (row entry, panel)
    // brings the whole bucket of a typical site.  A larger bucket keeps kInlineOverflowing entries inline and its
    // last word is the place of the others in `ent`.
    uint4 *inl = nullptr;
    uint32_t *ent = nullptr;      // overflow entries by bucket: record | nibble << 28 (any order inside a bucket)
    size_t inl_cap = 0, ent_cap = 0;
    uint32_t n_panels = 0;
    bool valid = false;
};

struct DeviceSet {
    uint4 *planes = nullptr;      // PL_COUNT * nchunks * npad
    uint32_t *counts = nullptr;   // npad x 4 {A,T,G,C}
    size_t planes_bytes = 0, counts_cap = 0;   // what the two allocations hold (bytes; records)
    size_t n = 0, len = 0, nchunks = 0, npad = 0;
    bool loaded = false;
    bool have_counts = false;
    // dst_upload_shared: only the planes of records [part_begin, part_end) are this rank's own work and valid; the lists
    // of ALL records came by the exchange.  Such a set runs on the consensus path only.
    bool partial = false;
    size_t part_begin = 0, part_end = 0;
    bool lean = false;            // only the four base planes are stored (a low-diversity set headed for the consensus path):
                                  // the dense pair kernels' other planes are derived on demand (ensure_derived)
    // the pack stored the base planes only of chunks that do not fit their slot (PackLists::defer_planes): the rest is the
    // reference plus rec.pre_slots, written by ensure_planes the first time anything but the consensus path reads planes
    bool planes_deferred = false;
    uint64_t epoch = 0;           // bumped by every upload: stale consensus indexes are rebuilt
    // consensus path
    ConsensusRef ref;
    RecordIndex rec;
    SiteIndex site;
    RunIndex runs;
    uint32_t *aconst = nullptr;   // [kMaxWords][npad] packed A_k words of one measure family (see the key below)
    size_t aconst_cap = 0;
    int aconst_family = -1;       // what `aconst` currently holds: family, packing, and the lists it was summed from
    bool aconst_wide = false;
    uint64_t aconst_epoch = 0, aconst_ref_epoch = 0;
    const void *aconst_ref_owner = nullptr;
    bool aconst_without_hot = false;
    // hybrid path: the hot columns of this set (by some set's reference) as a packed set of their own
    DeviceSet *hot = nullptr;
    const void *hot_ref_owner = nullptr;
    uint64_t hot_epoch = 0, hot_ref_epoch = 0;
};

struct BlockDesc {
    uint32_t i0, j0;  // first row / first column of the tile; i0 == 0xFFFFFFFF: idle filler
};

// geometry of one tile variant
struct TileShape {
    int bm;  // rows (scalar side) per block
    int bn;  // columns (vector side) per block = 256 * tn
};

struct PairLaunch {
    const DeviceSet *rows;
    const DeviceSet *cols;
    bool square;
    uint64_t row_begin, row_end;  // rows of `rows` this launch covers
    uint64_t out_base;            // canonical index of the first pair of this launch
    int out_kind;                 // DST_OUT_DISTANCE (f64 / int64 by measure), DST_OUT_TALLY or _TALLY16
    void *d_out;                  // first pair of this launch
    const BlockDesc *d_blocks;
    uint32_t nblocks;
    uint32_t ksplit = 1;          // > 1: split-L launch, partial tallies added atomically
};

struct ConsensusTile {
    uint32_t i0, i1, panel;  // rows [i0, i1) against column panel `panel`
};

// per-measure-family tables of the consensus path, built on the host (dst_host.cpp) from the per-site
// semantics of src/measures.rs and uploaded once per context
struct ConsensusLut {
    // [family][wide][ref class][row nibble][col nibble][word]   h_k packed like the accumulators
    // [family][wide][ref class][nibble][word]                    A-term of one difference site
    // [family][wide][word]                                       f_k(known, same known) = F per known ref site
    uint32_t h[4][2][kRefClasses][16][16][kMaxWords];
    uint32_t a[4][2][kRefClasses][16][kMaxWords];
    uint32_t unit[4][2][kMaxWords];
};
enum Family : int { FAM_NHIGH = 0, FAM_RAW = 1, FAM_K80 = 2, FAM_TN93 = 3 };
int family_of(int measure);
int family_words(int family, bool wide);       // accumulator words per pair
void build_consensus_lut(ConsensusLut &lut);   // dst_host.cpp
void pack_tallies(int family, bool wide, const int64_t tallies[4], uint32_t words[kMaxWords]);
// the per-site tallies of one code pair (bytes as src/encoding.rs produces them): the site loop bodies of
// src/measures.rs:14-23, 56-66, 85-107, 156-175.  out has tally_width(measure) entries.
void site_tallies(int measure, uint8_t q, uint8_t t, int out[4]);
std::vector<ConsensusTile> build_consensus_tiles(bool square, uint64_t row_begin, uint64_t row_end,
                                                 uint64_t n_cols, uint32_t rows_per_tile);

struct ConsensusLaunch {
    const DeviceSet *rows;
    const DeviceSet *cols;
    bool square;
    uint64_t row_begin, row_end, out_base;
    int out_kind;
    void *d_out;
    const ConsensusTile *d_tiles;
    uint32_t ntiles;
    bool wide;                   // one 32-bit word per tally (alignments of 65,536 sites or more)
    const ConsensusLut *d_lut;
    const void *d_hot = nullptr; // hybrid path: the dense kernels' tallies of the hot columns (TALLY16 / TALLY layout)
    int heavy_events = 0;        // 1: many events per pair or long lists — more of the workgroup's waves take the event role;
                                 // 2: more than one event per pair — every wave in both roles (consensus_pair_kernel)
};

// ---- shared preparation (dst_shared.cpp): one rank's exchange block, in 32-bit words ------------------
//   [0] entries of the block's lists   [1] 1: they did not fit ent_cap   [2, 3] first invalid byte (~0: none)   [4..15] 0
//   [cnt_at .. + rmax)        list lengths of the rank's records (record k * rmax + i of the set)
//   [counts_at .. + 4 rmax)   their {A,T,G,C} counts
//   [ent_at .. + ent_cap)     the entries, record after record
struct SharedLayout {
    uint32_t world, rmax, cnt_at, counts_at, ent_at, ent_cap, words;
};
SharedLayout shared_layout(uint64_t n, int world, uint32_t ent_cap);   // dst_host.cpp (pure host)

// ---- consensus-path launchers (dst_consensus.hip) --------------------------------------------
hipError_t launch_shared_block(uint32_t *block, const SharedLayout &lay, const DeviceSet &set, size_t rec_begin, size_t count,
                               const uint32_t *off_local, const unsigned long long *first_bad, hipStream_t stream);
// the gathered blocks -> set.rec.off / ent / range_start (and set.counts), then the report for the host (kReportWords + 1 words)
hipError_t launch_shared_splice(const uint32_t *gathered, const SharedLayout &lay, DeviceSet &set, uint32_t *len_all,
                                uint32_t *scan_tmp, bool with_counts, unsigned long long *report, hipStream_t stream);
hipError_t launch_ref_sample(const DeviceSet &set, hipStream_t stream);
// the same from the row-major code matrix (before the pack)
// ... on its way the kernel clears zero[0..zero_words) and sets *first_bad = ~0: what the pack behind it accumulates into
hipError_t launch_ref_sample_bytes(const uint8_t *d_codes, size_t row_stride, const DeviceSet &set, hipStream_t stream,
                                   uint32_t *zero = nullptr, size_t zero_words = 0, unsigned long long *first_bad = nullptr);
// count pass (fill == false): rec_cnt[n], site_cnt[len * n_panels] (when want_sites), *total
// fill pass: entries behind the scanned offsets.  ref_planes: [4][nchunks] uint4.
hipError_t launch_hot_list(const DeviceSet &set, hipStream_t stream);
hipError_t launch_compact(const DeviceSet &src, const uint32_t *hot_sites, uint32_t n_hot, DeviceSet &dst, hipStream_t stream);
// hot_planes != NULL: sites whose bit is set are left out of the lists
hipError_t launch_index(const DeviceSet &set, const uint4 *ref_planes, const uint4 *hot_planes, bool fill, bool skip_nclass,
                        uint32_t *rec_off_or_cnt, uint32_t *rec_ent, unsigned long long *total,
                        hipStream_t stream, uint32_t *range_start = nullptr);
// the same lists from the pack's slots (set.rec.pre_slots) instead of the planes; without_hot: leave the hot entries out
// rec_begin / rec_end / ent_cap: only those records (rec_off[0] = record rec_begin's offset), entries at or beyond ent_cap
// dropped — one rank's share of a set, written into an exchange block of fixed size (dst_upload_shared)
hipError_t launch_slot_fill(const DeviceSet &set, const uint4 *ref_planes, const uint4 *hot_planes, bool without_hot,
                            uint32_t *rec_off, uint32_t *rec_ent, uint32_t *range_start, hipStream_t stream, size_t rec_begin = 0,
                            size_t rec_end = ~(size_t)0, uint32_t ent_cap = 0xFFFFFFFFu);
// a column set's site buckets from its lists (set.rec -> set.site); *ovf_total (zeroed by the caller) counts the
// overflow entries placed in set.site.ent
hipError_t launch_site_buckets(const DeviceSet &set, uint32_t n_panels, uint32_t *ovf_total, hipStream_t stream);
// in-place exclusive scan of data[0..n) (data[n] receives the total); tmp: scan_tmp_words(n) words
size_t scan_tmp_words(size_t n);
// exclusive scan of data[0..n) in place; src0 (+ src1) given: of src0[i] (+ src1[i]) into data
// zero[0..n_zero) (n_zero <= 1024) is cleared before the scan's results are visible: counters the next kernels add to
hipError_t launch_exclusive_scan(uint32_t *data, size_t n, uint32_t *tmp, hipStream_t stream, const uint32_t *src0 = nullptr,
                                 const uint32_t *src1 = nullptr, uint32_t *zero = nullptr, uint32_t n_zero = 0);
hipError_t launch_aconst(const DeviceSet &set, int family, bool wide, const ConsensusLut *d_lut, hipStream_t stream);
// run records (RunIndex): the known reference sites per chunk (hot_planes != NULL: cold sites only), and the correction
// tables of one (family, packing) — after launch_aconst for the same, which leaves the entries' a-words in runs.aent
hipError_t launch_run_known(const DeviceSet &set, const uint4 *ref_planes, const uint4 *hot_planes, hipStream_t stream);
hipError_t launch_run_masks(const DeviceSet &set, hipStream_t stream);   // runs.mask from the slots' run-chunk flags
hipError_t launch_run_tables(const DeviceSet &set, int family, bool wide, bool without_hot, const ConsensusLut *d_lut, hipStream_t stream);
constexpr int kReportWords = 13;   // [0] first invalid byte, [1..8] the sample's statistics, [9..10] list totals, [11] run records, [12] entries their lists lost
// cnt_cold / cnt_hot (may be NULL): the pack's list lengths, summed into words 9 and 10.  runs: the pack's run-chunk
// counters — records with kRunMin run chunks and more become run records (numbered in runs->index / ids, word 11 = how
// many; 0 when there are more than max_run: stripping off), every other record gets its run chunks' entries back into
// its list length
// totals: four zeroed 64-bit words (the set's pre_totals block: the sample kernel clears it with the counters)
hipError_t launch_report(const unsigned long long *first_bad, const unsigned long long *stats, uint32_t *cnt_cold,
                         uint32_t *cnt_hot, size_t n, unsigned long long *report, hipStream_t stream, const RunIndex *runs = nullptr,
                         uint32_t max_run = 0, unsigned long long *totals = nullptr);
hipError_t launch_sum2_u32(const uint32_t *a0, const uint32_t *a1, size_t n, unsigned long long *totals, hipStream_t stream);
// f_words: F_k (known reference sites x the per-site unit), packed like the accumulators
hipError_t launch_consensus_pairs(int measure, const ConsensusLaunch &cl, const uint32_t f_words[kMaxWords],
                                  hipStream_t stream);
// exact per-site counts of known G, C, T over the records of `set`, added into hist[len][3]
hipError_t launch_site_hist(const DeviceSet &set, uint32_t *hist, hipStream_t stream);

const char *consensus_build_flags();   // dst_consensus.hip: the DST_DBG_* measurement macros it was compiled with

// ---- kernel launchers (dst_kernels.hip) -----------------------------------------------------
// lists != NULL: the pack also counts every record's differences from the reference (cold and hot sites apart)
struct PackLists {
    const uint4 *ref_planes, *hot_planes;      // [4][nchunks], [nchunks]
    const unsigned long long *stats;           // the sample's statistics (device): [1] = sum of deviants
    unsigned long long max_dev_sum;            // count only while stats[1] <= this (high-diversity sets go dense anyway)
    uint32_t *cnt_cold, *cnt_hot;              // [n], zeroed
    uint4 *slots;                              // [nchunks][npad] the differences of every (record, chunk), see pack_kernel
    // run chunks (RunIndex; all NULL: not looked for): counted apart and flagged in the slot
    uint32_t *cnt_run, *run_cold, *run_hot;
    // the base planes of a chunk that is inline in its slot are not stored (DeviceSet::planes_deferred)
    int defer_planes;
};
// a slot that holds every difference of its (record, chunk): not a chunk of N (flag 0x100), at most kSlotEntries of them
__host__ __device__ inline bool slot_is_inline(uint32_t word0) { return !(word0 & 0x100u) && (word0 & 0xFFu) <= kSlotEntries; }
// rec_begin / rec_end: only those records are packed (one rank's share of a set: dst_upload_shared)
hipError_t launch_pack(const uint8_t *d_codes, size_t row_stride, const DeviceSet &set,
                       unsigned long long *d_first_bad, const PackLists *lists, hipStream_t stream, size_t rec_begin = 0,
                       size_t rec_end = ~(size_t)0);
// the 4-bit wire format: rows of ceil(len / 2) bytes (row_stride a multiple of 64, base 16-byte aligned), two sites per
// byte, even site in the low nibble, a site = the high nibble of its Paradis code
hipError_t launch_pack_nibbles(const uint8_t *d_nibbles, size_t row_stride, const DeviceSet &set, unsigned long long *d_first_bad,
                               hipStream_t stream);
// lists: the pack's, when the set was packed with them (then a chunk that is inline in its slot is counted from the slot:
// its planes may not be there)
hipError_t launch_fill_counts(const DeviceSet &set, hipStream_t stream, const PackLists *lists = nullptr);
// the counts of records [rec_begin, rec_end) only, out[0..4) = record rec_begin's
hipError_t launch_range_counts(const DeviceSet &set, size_t rec_begin, size_t rec_end, uint32_t *out, hipStream_t stream,
                               const PackLists *lists = nullptr);
bool planes_deferred_by_pack();   // (false under DST_PACK_ALL_PLANES, the measurement knob)
// the base planes the pack deferred, from the slots and the reference (DeviceSet::planes_deferred)
hipError_t launch_planes_from_slots(const DeviceSet &set, hipStream_t stream);
hipError_t launch_derive(const DeviceSet &set, hipStream_t stream);   // planes K, X1, X0, CL of a lean set from its base planes
hipError_t launch_pairs(int measure, int variant, const PairLaunch &pl, hipStream_t stream);
// close: the reference's operation order with the table logarithm (what the text path uses), not the epilogue's arithmetic
hipError_t launch_finalize(int measure, const PairLaunch &pl, const void *d_tallies, bool tallies16,
                           void *d_out, hipStream_t stream, bool close = false);
TileShape tile_shape(int measure, int variant);
int variant_count(int measure);

// ---- host-only logic (dst_host.cpp) ----------------------------------------------------------
int tally_width(int measure);
bool measure_is_int(int measure);
// tiles of a launch, interleaved so that blocks b and b+8 (observed to share an XCD and its L2)
// walk the same column tile
std::vector<BlockDesc> build_blocks(bool square, uint64_t row_begin, uint64_t row_end,
                                    uint64_t n_cols, TileShape ts);
uint64_t square_row_start(uint64_t n, uint64_t i);
uint64_t pairs_in_rows(bool square, uint64_t n_cols, uint64_t row_begin, uint64_t row_end);

}  // namespace dst
