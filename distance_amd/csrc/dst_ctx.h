// dst_ctx.h — the context behind the C ABI and the helpers its translation units share
// (dst_api.cpp: uploads and runs, dst_stream.cpp: the stream-mode pipeline, dst_gather.cpp: multi-GPU gather).
#pragma once
#include <string>
#include <vector>

#include "dst_internal.h"

using namespace dst;

struct dst_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    DeviceSet set[2];
    // staging for host uploads / unaligned device inputs
    uint8_t *stage = nullptr;
    size_t stage_bytes = 0;
    unsigned long long *d_first_bad = nullptr;
    // what an upload reports to the host (first invalid byte, the sample's statistics, the list totals): written by the
    // device into this page-locked block, so the upload's one synchronisation needs no device-to-host copy
    unsigned long long *h_report = nullptr, *d_report = nullptr;
    // tile schedules already on the device, keyed by the launch geometry (multi-GPU runs cycle
    // through a few sub-slab ranges every step: no host sync or H2D on a hit)
    struct Schedule {
        bool square = false;
        uint64_t rb = 0, re = 0, ncols = 0;
        int bm = 0, bn = 0;  // dense tile shape; consensus-path tile lists: bm = rows per tile, bn = -1
        uint32_t nblocks = 0;
        void *d_blocks = nullptr;
        size_t bytes = 0;  // capacity of d_blocks
        uint64_t last_use = 0;
        std::vector<hipStream_t> users;   // the streams whose launches read it (waited for before its buffer is recycled)
    };
    std::vector<Schedule> schedules;
    uint64_t schedule_clock = 0;
    int variant = 0;
    int path = DST_PATH_AUTO;         // dst_set_path
    double prep_min_work = 2.0e10;    // dst_set_prep_threshold: site comparisons below which DST_PATH_AUTO stays dense unasked
    int last_path = DST_PATH_DENSE;   // what the most recent run used
    // consensus path: tables, counters and scratch shared by the two sets
    ConsensusLut *d_lut = nullptr;
    unsigned long long *d_total = nullptr;
    uint32_t *scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    void *hot_tally = nullptr;  // hybrid path: the dense kernels' tallies of the hot columns (grow-only)
    size_t hot_tally_bytes = 0;
    hipEvent_t hot_free = nullptr;  // recorded after the last reader of `hot_tally`
    bool hot_used = false;
    // cross-stream ordering of derived data (dst_api.cpp: publish_prep / order_after_prep / wait_for_other_runs)
    hipEvent_t prep_event = nullptr;
    hipStream_t prep_stream = nullptr;
    bool prep_pending = false;
    struct Recent {
        hipStream_t stream = nullptr;
        hipEvent_t event = nullptr;
        bool used = false;
    } recent[4];
    unsigned recent_next = 0;
    void *host_out = nullptr;  // device staging of the *_host run forms (grow-only)
    size_t host_out_bytes = 0;
    int ksplit = 0;  // 0 = automatic split-L factor, >= 1 forced
    uint32_t *scratch = nullptr;  // partial-tally meeting buffer of split-L f64 runs
    size_t scratch_bytes = 0;
    hipEvent_t scratch_free = nullptr;  // recorded after the last reader of `scratch`
    bool scratch_used = false;
    // TSV text on the device (dst_text.hip): the sets' record ids and grow-only scratch
    struct Ids {
        uint32_t *off = nullptr;  // [n + 1] into chars
        char *chars = nullptr;
        size_t off_bytes = 0, chars_bytes = 0;
        uint64_t n = 0;
    } ids[2];
    void *text_res = nullptr;      // the slab's results (8 B per pair) or tallies (<= 16 B per pair)
    void *text_num = nullptr;      // 32-byte number records
    uint32_t *text_len = nullptr;  // line lengths -> offsets
    uint32_t *text_scan = nullptr;
    char *text_buf = nullptr;
    uint32_t *text_flag = nullptr;   // [0] a value without a short text, [1] near ties noted
    void *text_ties = nullptr;       // the slab's near ties (dst_text.hip: NearTie), device and page-locked host copies
    void *text_ties_host = nullptr;
    size_t text_res_bytes = 0, text_num_bytes = 0, text_len_bytes = 0, text_scan_bytes = 0, text_buf_bytes = 0;
    size_t text_ties_bytes = 0, text_ties_host_bytes = 0;
    // the sets' {A,T,G,C} counts on the host (tn93 near ties are re-finalised there), valid while the epoch matches
    std::vector<uint32_t> text_counts[2];
    uint64_t text_counts_epoch[2] = {~0ull, ~0ull};
    uint64_t text_near_ties = 0, text_patched = 0;   // running totals (dst_text_stats)
    // dst_upload_shared (dst_shared.cpp): this rank's exchange block, everybody's blocks, and what the last exchange told
    struct Shared {
        void *send = nullptr, *recv = nullptr;
        uint32_t *off_local = nullptr;
        size_t send_bytes = 0, recv_bytes = 0, off_local_bytes = 0;
        uint64_t last_biggest = 0;   // entries of the largest block of the previous shared upload (sizes the next one)
        uint64_t uploads = 0, fallbacks = 0;
    } shared[2];
    // HIP events around the pair kernel ([0]) and the pack kernel ([1]) of the most recent launches, recorded on the launch
    // stream: a ring, so that a caller timing many steps reads them ONCE at the end (dst_kernel_ms_mean) instead of
    // waiting for the device after every step
    static constexpr int kTimerRing = 64;
    struct Timer {
        hipEvent_t begin[kTimerRing] = {}, end[kTimerRing] = {};
        uint64_t seq = 0, mark = 0;   // launches timed so far; where the running mean starts
    } timer[2];
    float pair_ms = 0, pack_ms = 0;
    bool timed_pair = false, timed_pack = false;
    std::string err;
};

struct dst_comm;

namespace dst {

// Lists are only worth counting while the sampled records deviate from the reference at less than this share of the
// sites (unstructured data crosses over to the dense path near 3-4 %; profiles/r02/consensus_calibration.txt)
constexpr double kListsMaxDeviation = 0.08;

int fail(dst_ctx *ctx, int status, const std::string &msg);
// bracket the next pair (which = 0) / pack (1) kernel on `stream` with the timer ring's events
int timer_begin(dst_ctx *ctx, int which, hipStream_t stream);
int timer_end(dst_ctx *ctx, int which, hipStream_t stream);
int fail_hip(dst_ctx *ctx, hipError_t e, const char *what);

#define HIP_TRY(ctx, call)                       \
    do {                                         \
        hipError_t e_ = (call);                  \
        if (e_ != hipSuccess)                    \
            return dst::fail_hip((ctx), e_, #call); \
    } while (0)

int ensure_bytes(dst_ctx *ctx, void **ptr, size_t *have, size_t want);
void free_set(DeviceSet &s);
// queue the pack of an n x len byte matrix (device memory) into `s`; *d_first_bad receives the index of the first
// byte that is not a Paradis code (or stays ~0).  Nothing here waits for the device.
// nibbles: d_codes holds the 4-bit wire format (two sites per byte) instead of Paradis bytes
int pack_queue(dst_ctx *ctx, DeviceSet &s, const uint8_t *d_codes, size_t n, size_t len, size_t row_stride,
               const uint32_t *d_counts, unsigned long long *d_first_bad, hipStream_t stream, bool want_lists, bool nibbles = false);
int invalid_code_error(dst_ctx *ctx, unsigned long long first_bad, size_t len);
// the per-record {A,T,G,C} counts of `s` on the device (counted by code unless the upload brought them)
int need_counts(dst_ctx *ctx, DeviceSet &s, hipStream_t stream);
// rows [rb, re) of `rows` against every (square: later) record of `cols` — any two packed sets of this context
int run_sets(dst_ctx *ctx, int measure, bool square, DeviceSet &rows, DeviceSet &cols, uint64_t rb, uint64_t re,
             int out_kind, void *d_out, size_t cap, void *stream_v);

}  // namespace dst
