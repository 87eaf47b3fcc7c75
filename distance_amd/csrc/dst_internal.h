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

struct DeviceSet {
    uint4 *planes = nullptr;      // PL_COUNT * nchunks * npad
    uint32_t *counts = nullptr;   // npad x 4 {A,T,G,C}
    size_t planes_bytes = 0;
    size_t n = 0, len = 0, nchunks = 0, npad = 0;
    bool loaded = false;
    bool have_counts = false;
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

// ---- kernel launchers (dst_kernels.hip) -----------------------------------------------------
hipError_t launch_pack(const uint8_t *d_codes, size_t row_stride, const DeviceSet &set,
                       unsigned long long *d_first_bad, hipStream_t stream);
hipError_t launch_fill_counts(const DeviceSet &set, hipStream_t stream);
hipError_t launch_pairs(int measure, int variant, const PairLaunch &pl, hipStream_t stream);
hipError_t launch_finalize(int measure, const PairLaunch &pl, const void *d_tallies, bool tallies16,
                           void *d_out, hipStream_t stream);
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
