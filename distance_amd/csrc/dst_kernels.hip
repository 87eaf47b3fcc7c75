// dst_kernels.hip — hand-written gfx950 (MI355X, CDNA4) kernels of the all-pairs distance path.
//
//   pack_kernel      Paradis bytes (src/encoding.rs:4-41)  ->  8 bit-planes, validates codes
//   counts_kernel    per-record {A,T,G,C} counts            (src/fastaio.rs:53-66)
//   pair_kernel<M>   site tallies of every pair of a tile   (src/measures.rs:14-23, 56-66,
//                                                            85-107, 156-175)
//                    and, fused in its epilogue, tallies -> f64 distance in the reference's
//                    operation order              (src/measures.rs:68, 76, 109-112, 118-190)
//
//   finalize_kernel  tallies already in HBM -> distances (split-L launches, dst_finalize_device)
//
// Design (DESIGN.md has the numbers): integer / bitwise work, no MFMA.  One pair costs 5..8 VALU
// ops per 32 sites (v_and_b32, v_bitop3_b32, v_bcnt_u32_b32).  A block owns BM "row" records x
// 256*TN "column" records.  Each lane owns TN column records and keeps their plane words of the
// current 128-site chunk in VGPRs; the BM row records of the tile are wave-uniform and come from
// a small double-buffered LDS tile by broadcast ds_read_b128 — NOT through scalar loads: on gfx950
// a VALU op with an SGPR operand issues at half rate (tools/ubench/valu_rate.hip).  Tallies live
// in VGPRs for the whole sweep over L.  Tiles of one column panel are dealt to workgroups that
// share an XCD so the panel is served by that XCD's L2.  The kernel runs at the measured issue
// ceiling of its instruction mix (tools/ubench/ifetch.hip, order.hip).
#include <algorithm>

#include "dst_device.hpp"

namespace dst {

// =============================================================================================
// pack: bytes -> planes
// =============================================================================================
namespace {

// Bit BP of each of the 32 bytes x[0..7] -> one 32-bit plane word (byte k of the group -> bit k).  The gather is two
// v_dot4_u32_u8 per 8 bytes: the masked bytes are 0 or 2^BP, the weights 1,2,4,8 / 16,32,64,128 put them on distinct
// bits (sum <= 255 * 2^BP: no carry).  (r01/r02 used a 32-bit multiply per 4 bytes and plane — quarter rate on CDNA —
// for each of the 8 planes: 360 M VALU instructions per 50,000 x 30,000 pack, 0.59 ms of issue; now the four base planes
// and the four low code bits are gathered and everything else is word-level logic.)
template <int BP>
__device__ __forceinline__ uint32_t gather32(const uint32_t (&x)[8])
{
    constexpr uint32_t M = 0x01010101u << BP;
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t r = __builtin_amdgcn_udot4(x[2 * k + 1] & M, 0x80402010u,
                                                  __builtin_amdgcn_udot4(x[2 * k] & M, 0x08040201u, 0u, false), false);
        w |= (r >> BP) << (8 * k);
    }
    return w;
}

// K, X1, X0, CL of 32 sites from their base planes (for the 17 valid codes K is code bit 3: exactly one base)
__device__ __forceinline__ void derive_planes(uint32_t A, uint32_t G, uint32_t C, uint32_t T, uint32_t &K, uint32_t &X1,
                                              uint32_t &X0, uint32_t &CL)
{
    const uint32_t pur = A | G, pyr = C | T;
    K = (A ^ G ^ C ^ T) & ~((A & G) | (C & T));   // exactly one base
    X1 = pyr & ~pur;                               // {C,T,Y}
    CL = X1 | (pur & ~pyr);                        // plus {A,G,R}
    X0 = K & (G | T);
}

// One thread = one (record, 128-site chunk).  Lanes run along records, so the eight 16-byte
// plane stores of a wave are 1 KiB contiguous each; every lane reads its own 128-byte line.
// Sites >= len and records >= n are filled with N (0xF0): N contributes nothing to any tally.
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t *__restrict__ codes,
                                                   size_t row_stride, uint32_t n, uint32_t len,
                                                   uint32_t nchunks, uint32_t npad,
                                                   uint4 *__restrict__ planes,
                                                   unsigned long long *__restrict__ first_bad,
                                                   int aligned16, PackLists lists, uint32_t rec_first, uint32_t rec_last)
{
    // records [rec_first, rec_last): the whole padded set, or one rank's share of it (dst_upload_shared)
    const uint32_t s = rec_first + blockIdx.y * blockDim.x + threadIdx.x;
    const uint32_t c = blockIdx.x;
    if (s >= rec_last)
        return;
    uint32_t out[PL_COUNT][4];
#pragma unroll
    for (int p = 0; p < PL_COUNT; ++p)
#pragma unroll
        for (int w = 0; w < 4; ++w)
            out[p][w] = (p <= PL_T) ? 0xFFFFFFFFu : 0u;  // all N

    if (s < n) {
        const uint32_t site0 = c * kChunkSites;
        const uint8_t *row = codes + (size_t)s * row_stride + site0;
        const bool fast = (aligned16 & 1) && site0 + kChunkSites <= len;
        // the shifted path reads from 3 bytes before to 4 bytes after the chunk's 128: inside the matrix unless this
        // is its very first chunk (and the matrix starts off a 4-byte boundary) or within 132 bytes of its end
        const bool words = !fast && (size_t)s * row_stride + site0 + kChunkSites + 4 <= (size_t)(n - 1) * row_stride + len &&
                           (s > 0 || c > 0 || (aligned16 & 2));
        uint32_t bad_at = 0xFFFFFFFFu;
        // the lane's whole 128-byte line at once: its eight 16-byte loads are in flight together, so the line is
        // consumed while it is still in the L1 (spread over the four words below it was fetched again and again)
        uint4 in[8];
        if (fast) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                in[k] = reinterpret_cast<const uint4 *>(row)[k];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("" : "+v"(in[k].x), "+v"(in[k].y), "+v"(in[k].z), "+v"(in[k].w));
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t x[8];
            if (fast) {
                const uint4 lo = in[2 * w], hi = in[2 * w + 1];
                x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w;
                x[4] = hi.x; x[5] = hi.y; x[6] = hi.z; x[7] = hi.w;
            } else if (words) {
                // rows on any boundary (29,903- or 1,000-site records): aligned 4-byte loads around the 32 bytes and a
                // funnel shift (v_alignbit) — byte loads cost 25x the aligned path
                const uintptr_t a = reinterpret_cast<uintptr_t>(row) + 32u * w;
                const uint32_t *p = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
                const uint32_t sh = (uint32_t)(a & 3u) * 8u;
                uint32_t d[9];
#pragma unroll
                for (int g = 0; g < 9; ++g)
                    d[g] = p[g];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    x[g] = __builtin_amdgcn_alignbit(d[g + 1], d[g], sh);
                    // the last chunk of a row: what lies past the row's end (the next row's bytes) becomes N
                    const uint32_t first = site0 + 32u * w + 4u * g;
                    if (first + 4 > len) {
                        const uint32_t keep = first < len ? len - first : 0u;   // bytes of this word inside the row
                        const uint32_t m = keep ? 0xFFFFFFFFu >> (32u - 8u * keep) : 0u;
                        x[g] = (x[g] & m) | (0xF0F0F0F0u & ~m);
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    uint32_t v = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const uint32_t off = 32 * w + 4 * g + b;
                        const uint32_t byte = site0 + off < len ? row[off] : 0xF0u;
                        v |= byte << (8 * b);
                    }
                    x[g] = v;
                }
            }
            const uint32_t A = gather32<7>(x), G = gather32<6>(x), C = gather32<5>(x), T = gather32<4>(x);
            uint32_t K, X1, X0, CL;
            derive_planes(A, G, C, T, K, X1, X0, CL);
            // validity (src/encoding.rs produces 17 codes): one base bit -> low nibble 8; two or three -> 0;
            // all four -> 0 (N), 4 (-) or 2 (?); none -> not a code
            const uint32_t l3 = gather32<3>(x), l2 = gather32<2>(x), l1 = gather32<1>(x), l0 = gather32<0>(x);
            const uint32_t all4 = A & G & C & T, any = A | G | C | T, low0 = ~(l3 | l2 | l1 | l0);
            const uint32_t ok = (K & l3 & ~(l2 | l1 | l0)) | (any & ~K & ~all4 & low0) | (all4 & ~l3 & ~l0 & ~(l2 & l1));
            const uint32_t bad = ~ok;
            if (bad && bad_at == 0xFFFFFFFFu)
                bad_at = 32 * w + (uint32_t)__builtin_ctz(bad);
            out[PL_A][w] = A;
            out[PL_G][w] = G;
            out[PL_C][w] = C;
            out[PL_T][w] = T;
            out[PL_K][w] = K;
            out[PL_X1][w] = X1;
            out[PL_X0][w] = X0;
            out[PL_CL][w] = CL;
        }
        if (bad_at != 0xFFFFFFFFu)
            atomicMin(first_bad, (unsigned long long)s * len + site0 + bad_at);
    }
    // The consensus path's first pass, for free: how many sites of this (record, chunk) differ from the reference
    // sequence (sampled from the bytes before this kernel).  Cold and hot sites apart: the hybrid path leaves the hot
    // ones out of the lists.  Skipped when the sample says the set is too diverse for lists.
    const bool low_diversity = lists.ref_planes && lists.stats[1] <= lists.max_dev_sum;
    bool inline_slot = false;   // the slot says all there is to say about this chunk: its planes are not stored
    if (low_diversity) {
        uint32_t cold = 0, hot = 0;
        // ... and the entries themselves, while the codes are in registers: one 16-byte slot per (record, chunk) —
        // halfword 0 = differences in the chunk, then up to kSlotEntries of them in site order as
        // site-in-chunk | class of the reference << 7 | nibble << 10 | hot << 14.  The fill pass reads these slots
        // (1/4 of the planes it would read) and goes back to the planes only for a chunk with more differences.
        uint32_t slot[4] = {0, 0, 0, 0};
        inline_slot = lists.defer_planes != 0;   // (a padding record: all N, nothing to keep)
        if (s < n) {
            const uint4 h4 = lists.hot_planes[c];
            const uint32_t hw[4] = {h4.x, h4.y, h4.z, h4.w};
            uint32_t rw[4][4];
#pragma unroll
            for (int p = 0; p <= PL_T; ++p) {
                const uint4 r4 = lists.ref_planes[(size_t)p * nchunks + c];
                rw[p][0] = r4.x; rw[p][1] = r4.y; rw[p][2] = r4.z; rw[p][3] = r4.w;
            }
            uint32_t d[4], k = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                d[w] = (out[PL_A][w] ^ rw[PL_A][w]) | (out[PL_G][w] ^ rw[PL_G][w]) | (out[PL_C][w] ^ rw[PL_C][w]) |
                       (out[PL_T][w] ^ rw[PL_T][w]);
                cold += __builtin_popcount(d[w] & ~hw[w]);
                hot += __builtin_popcount(d[w] & hw[w]);
            }
            // a run chunk: 128 sites of N (N, -, ?) where the reference has a known base — its entries are counted apart,
            // the slot notes it: whether they stay in the list is decided once the record's run
            // chunks are counted (finish_kernel, slot_fill_kernel; dst_internal.h: RunIndex)
            bool run_chunk = false;
            if (lists.cnt_run && cold + hot) {
                uint32_t all_n = 0xFFFFFFFFu;
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    all_n &= out[PL_A][w] & out[PL_G][w] & out[PL_C][w] & out[PL_T][w];
                run_chunk = all_n == 0xFFFFFFFFu;   // (cold + hot != 0: the reference is not all N-class here)
            }
            if (run_chunk) {
                atomicAdd(&lists.cnt_run[s], 1u);
                if (cold)
                    atomicAdd(&lists.run_cold[s], cold);
                if (hot)
                    atomicAdd(&lists.run_hot[s], hot);
                slot[0] = min(cold + hot, 255u) | 0x100u;
            } else if (cold + hot) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    uint32_t m = d[w];
                    while (m && k < kSlotEntries) {
                        const uint32_t bit = (uint32_t)__builtin_ctz(m);
                        m &= m - 1;
                        const uint32_t nib = ((out[PL_A][w] >> bit) & 1u) << 3 | ((out[PL_G][w] >> bit) & 1u) << 2 |
                                             ((out[PL_C][w] >> bit) & 1u) << 1 | ((out[PL_T][w] >> bit) & 1u);
                        const uint32_t rnib = ((rw[PL_A][w] >> bit) & 1u) << 3 | ((rw[PL_G][w] >> bit) & 1u) << 2 |
                                              ((rw[PL_C][w] >> bit) & 1u) << 1 | ((rw[PL_T][w] >> bit) & 1u);
                        const uint32_t cls = rnib == 8 ? 0u : rnib == 4 ? 1u : rnib == 2 ? 2u : rnib == 1 ? 3u : 4u;
                        const uint32_t e = (32u * w + bit) | cls << 7 | nib << 10 | ((hw[w] >> bit) & 1u) << 14;
                        ++k;   // halfword k of the slot
                        slot[k >> 1] |= e << (16u * (k & 1u));
                    }
                }
                slot[0] |= min(cold + hot, 255u);
                atomicAdd(&lists.cnt_cold[s], cold);   // (adding 0 for an all-hot chunk is harmless)
                if (hot)
                    atomicAdd(&lists.cnt_hot[s], hot);
            }
            inline_slot = inline_slot && slot_is_inline(slot[0]);
        }
        lists.slots[(size_t)c * npad + s] = make_uint4(slot[0], slot[1], slot[2], slot[3]);
    }
    // A set headed for the consensus path is stored lean: the K, X1, X0 and CL planes are functions of the four base
    // planes (derive_planes below) that only the dense pair kernels read; derive_kernel builds them if one ever runs.
    // And its base planes are deferred (PackLists::defer_planes): the consensus path reads lists, not planes, and a chunk
    // whose differences all sit in its slot is the reference plus those — planes_from_slots_kernel writes it if a dense
    // kernel ever asks (half of this kernel's HBM traffic was these stores: 0.75 of 2.5 GB at 50,000 x 30,000).  Only a
    // chunk that does not fit its slot (more than kSlotEntries differences, a chunk of N) is stored now: the fill pass
    // reads those from the planes.
#pragma unroll
    for (int p = 0; p < PL_COUNT; ++p)
        if (p <= PL_T ? !inline_slot : !low_diversity)
            planes[((size_t)p * nchunks + c) * npad + s] = make_uint4(out[p][0], out[p][1], out[p][2], out[p][3]);
}

// The same from the 4-bit wire format of the stream pipeline (dst_stream_open_wire, DST_WIRE_NIBBLES): a site is the
// high nibble of its Paradis code — all that any measure reads (bit 3 of the code, "known", is "exactly one base") —
// two sites per byte, the even site in the low nibble.  Half the bytes over the host link, which is what bounds a
// streamed job (1,000 x 5 Mbp loaded vs streamed batches: 320 MB per 64-record batch at 8 bits).  One thread = one
// (record, 128-site chunk) = 64 bytes of input; nibble 0 (no base) is not a code.
template <int BIT>   // plane bit within the nibble: 3 = A, 2 = G, 1 = C, 0 = T
__device__ __forceinline__ uint32_t gather_nibbles(uint32_t x)   // 8 sites of one 32-bit word -> 8 plane bits
{
    const uint32_t even = __builtin_amdgcn_udot4((x >> BIT) & 0x01010101u, 0x40100401u, 0u, false);
    return __builtin_amdgcn_udot4((x >> (BIT + 4)) & 0x01010101u, 0x80200802u, even, false);
}

__global__ __launch_bounds__(256) void pack_nibbles_kernel(const uint8_t *__restrict__ nibbles, size_t row_stride, uint32_t n,
                                                           uint32_t len, uint32_t nchunks, uint32_t npad,
                                                           uint4 *__restrict__ planes,
                                                           unsigned long long *__restrict__ first_bad)
{
    const uint32_t s = blockIdx.y * blockDim.x + threadIdx.x;
    const uint32_t c = blockIdx.x;
    if (s >= npad)
        return;
    uint32_t out[PL_COUNT][4];
#pragma unroll
    for (int p = 0; p < PL_COUNT; ++p)
#pragma unroll
        for (int w = 0; w < 4; ++w)
            out[p][w] = (p <= PL_T) ? 0xFFFFFFFFu : 0u;  // all N
    if (s < n) {
        const uint32_t site0 = c * kChunkSites;
        const uint4 *row = reinterpret_cast<const uint4 *>(nibbles + (size_t)s * row_stride + site0 / 2);
        uint4 in[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            in[k] = row[k];
        uint32_t bad_at = 0xFFFFFFFFu;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t x[4] = {in[w].x, in[w].y, in[w].z, in[w].w};
            uint32_t A = 0, G = 0, C = 0, T = 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint32_t first = site0 + 32u * w + 8u * g;
                if (first + 8 > len) {   // the row's last word: what lies past its end becomes N
                    const uint32_t keep = first < len ? len - first : 0u;
                    const uint32_t m = keep ? 0xFFFFFFFFu >> (32u - 4u * keep) : 0u;
                    x[g] = (x[g] & m) | ~m;
                }
                const uint32_t zero = (x[g] - 0x11111111u) & ~x[g] & 0x88888888u;   // some nibble is 0 (the lowest mark is exact)
                if (zero && bad_at == 0xFFFFFFFFu)
                    bad_at = 32u * w + 8u * g + ((uint32_t)__builtin_ctz(zero) >> 2);
                A |= gather_nibbles<3>(x[g]) << (8 * g);
                G |= gather_nibbles<2>(x[g]) << (8 * g);
                C |= gather_nibbles<1>(x[g]) << (8 * g);
                T |= gather_nibbles<0>(x[g]) << (8 * g);
            }
            out[PL_A][w] = A;
            out[PL_G][w] = G;
            out[PL_C][w] = C;
            out[PL_T][w] = T;
            derive_planes(A, G, C, T, out[PL_K][w], out[PL_X1][w], out[PL_X0][w], out[PL_CL][w]);
        }
        if (bad_at != 0xFFFFFFFFu)
            atomicMin(first_bad, (unsigned long long)s * len + site0 + bad_at);
    }
#pragma unroll
    for (int p = 0; p < PL_COUNT; ++p)
        planes[((size_t)p * nchunks + c) * npad + s] = make_uint4(out[p][0], out[p][1], out[p][2], out[p][3]);
}

// the four derived planes of a set that was packed lean
__global__ __launch_bounds__(256) void derive_kernel(uint4 *__restrict__ planes, size_t per_plane)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_plane)
        return;
    const uint4 A = planes[PL_A * per_plane + i], G = planes[PL_G * per_plane + i], C = planes[PL_C * per_plane + i],
                T = planes[PL_T * per_plane + i];
    uint4 K, X1, X0, CL;
    derive_planes(A.x, G.x, C.x, T.x, K.x, X1.x, X0.x, CL.x);
    derive_planes(A.y, G.y, C.y, T.y, K.y, X1.y, X0.y, CL.y);
    derive_planes(A.z, G.z, C.z, T.z, K.z, X1.z, X0.z, CL.z);
    derive_planes(A.w, G.w, C.w, T.w, K.w, X1.w, X0.w, CL.w);
    planes[PL_K * per_plane + i] = K;
    planes[PL_X1 * per_plane + i] = X1;
    planes[PL_X0 * per_plane + i] = X0;
    planes[PL_CL * per_plane + i] = CL;
}

// The base planes of a set whose pack deferred them (DeviceSet::planes_deferred): one thread = one (record, chunk), as in
// the pack.  A chunk that is inline in its slot is the reference's planes with the slot's entries written over them (an
// entry = site in the chunk | ... | the record's four base bits << 10); the others the pack stored itself; padding
// records are N.
__global__ __launch_bounds__(256) void planes_from_slots_kernel(const uint4 *__restrict__ slots, const uint4 *__restrict__ ref_planes,
                                                                uint32_t n, uint32_t nchunks, uint32_t npad,
                                                                uint4 *__restrict__ planes)
{
    const uint32_t s = blockIdx.y * blockDim.x + threadIdx.x;
    const uint32_t c = blockIdx.x;
    if (s >= npad)
        return;
    uint32_t out[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int w = 0; w < 4; ++w)
            out[p][w] = 0xFFFFFFFFu;   // all N
    if (s < n) {
        const uint4 slot = slots[(size_t)c * npad + s];
        if (!slot_is_inline(slot.x))
            return;
        const uint32_t sw[4] = {slot.x, slot.y, slot.z, slot.w};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint4 r4 = ref_planes[(size_t)p * nchunks + c];
            out[p][0] = r4.x; out[p][1] = r4.y; out[p][2] = r4.z; out[p][3] = r4.w;
        }
        const uint32_t cnt = slot.x & 0xFFu;
#pragma unroll
        for (uint32_t k = 1; k <= kSlotEntries; ++k) {
            const uint32_t e = (sw[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu;
            const uint32_t bit = k <= cnt ? 1u << (e & 31u) : 0u, word = (e >> 5) & 3u, nib = (e >> 10) & 15u;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t m = word == (uint32_t)w ? bit : 0u;
#pragma unroll
                for (int p = 0; p < 4; ++p)   // planes A, G, C, T = nibble bits 3, 2, 1, 0
                    out[p][w] = (out[p][w] & ~m) | (((nib >> (3 - p)) & 1u) ? m : 0u);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
        planes[((size_t)p * nchunks + c) * npad + s] = make_uint4(out[p][0], out[p][1], out[p][2], out[p][3]);
}

__device__ __forceinline__ void count_known(uint32_t A, uint32_t G, uint32_t C, uint32_t T, int &a, int &t, int &g, int &cc)
{
    const uint32_t K = (A ^ G ^ C ^ T) & ~((A & G) | (C & T));   // exactly one base (a lean set has no K plane)
    a += __builtin_popcount(K & A);
    t += __builtin_popcount(K & T);
    g += __builtin_popcount(K & G);
    cc += __builtin_popcount(K & C);
}
__device__ __forceinline__ void count_known4(const uint4 &A, const uint4 &G, const uint4 &C, const uint4 &T, int &a, int &t, int &g,
                                             int &cc)
{
    count_known(A.x, G.x, C.x, T.x, a, t, g, cc);
    count_known(A.y, G.y, C.y, T.y, a, t, g, cc);
    count_known(A.z, G.z, C.z, T.z, a, t, g, cc);
    count_known(A.w, G.w, C.w, T.w, a, t, g, cc);
}

// {A,T,G,C} counts by code (src/fastaio.rs:53-66): a known base is K & its own bit-plane.  A wave = 8 records x 8 chunk
// lanes (128 contiguous bytes per chunk and plane).  With slots (a set packed with its lists): a chunk that is inline in
// its slot has no planes yet — its counts are the reference's chunk, less the reference's base and plus the record's at
// every entry of the slot (a quarter of the planes' bytes).
__global__ __launch_bounds__(256) void counts_kernel(const uint4 *__restrict__ planes, const uint4 *__restrict__ slots,
                                                     const uint4 *__restrict__ ref_planes,
                                                     const unsigned long long *__restrict__ stats, unsigned long long max_dev_sum,
                                                     uint32_t n, uint32_t nchunks, uint32_t npad,
                                                     uint32_t *__restrict__ counts, uint32_t rec_first, uint32_t rec_last)
{
    // counts[0] is record rec_first's (the whole padded set, or one rank's share of it)
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t rl = lane & 7u, cl = lane >> 3;
    const uint32_t s = rec_first + wave * 8u + rl;
    const bool deferred = slots && (!stats || stats[1] <= max_dev_sum);   // (stats: the pack's own test, when the host has not seen it yet)
    int a = 0, t = 0, g = 0, cc = 0;
    const size_t ps = (size_t)nchunks * npad;
    if (s < rec_last && s < n) {
        for (uint32_t c = cl; c < nchunks; c += 8u) {
            const size_t at = (size_t)c * npad + s;
            uint4 slot = make_uint4(0x100u, 0, 0, 0);
            if (deferred)
                slot = slots[at];
            if (slot_is_inline(slot.x)) {
                count_known4(ref_planes[c], ref_planes[nchunks + c], ref_planes[2 * (size_t)nchunks + c],
                             ref_planes[3 * (size_t)nchunks + c], a, t, g, cc);
                const uint32_t sw[4] = {slot.x, slot.y, slot.z, slot.w}, cnt = slot.x & 0xFFu;
#pragma unroll
                for (uint32_t k = 1; k <= kSlotEntries; ++k) {
                    const uint32_t e = (sw[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu;
                    if (k <= cnt) {
                        const uint32_t cls = (e >> 7) & 7u, nib = (e >> 10) & 15u;   // reference: 0 A, 1 G, 2 C, 3 T, 4 none
                        a += (int)(nib == 8u) - (int)(cls == 0u);
                        g += (int)(nib == 4u) - (int)(cls == 1u);
                        cc += (int)(nib == 2u) - (int)(cls == 2u);
                        t += (int)(nib == 1u) - (int)(cls == 3u);
                    }
                }
            } else {
                count_known4(planes[PL_A * ps + at], planes[PL_G * ps + at], planes[PL_C * ps + at], planes[PL_T * ps + at],
                             a, t, g, cc);
            }
        }
    }
#pragma unroll
    for (uint32_t o = 8; o < 64u; o <<= 1) {
        a += __shfl_xor(a, o);
        t += __shfl_xor(t, o);
        g += __shfl_xor(g, o);
        cc += __shfl_xor(cc, o);
    }
    if (cl == 0 && s < rec_last)
        reinterpret_cast<uint4 *>(counts)[s - rec_first] = make_uint4((uint32_t)a, (uint32_t)t, (uint32_t)g, (uint32_t)cc);
}

// =============================================================================================
// measures: per-32-site step on plane words + conversion of the raw popcounts to the
// reference's tallies.  q = row record (SGPR operands), t = column record (VGPR operands).
// =============================================================================================
// The inner ops are pinned.  Left to itself hipcc rebalances the OR-of-ANDs into 5 ops instead
// of 4 and splits chained popcount-accumulates into v_bcnt(x,0) + v_add3 (1.5 issue slots per
// tally instead of 1).  Boolean steps use the gfx950 v_bitop3_b32 builtin (any function of three
// inputs, TT = f(0xF0, 0xCC, 0xAA)); the accumulate is a one-instruction asm.  Only the bcnt is
// asm: hipcc pads back-to-back dependent asm statements with s_nop (dst-forwarding hazard it
// cannot rule out), and with ordinary VALU in between the scheduler never needs to.
template <int TT>
__device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
}
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t b, uint32_t c)  // (a & b) | c
{
    return bitop3<((0xF0 & 0xCC) | 0xAA)>(a, b, c);
}
__device__ __forceinline__ void bcnt_acc(uint32_t &acc, uint32_t x)  // acc += popcount(x)
{
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x));
}
constexpr int TT_AND3 = 0xF0 & 0xCC & 0xAA;          // a & b & c
constexpr int TT_A_AND_B_EQ_C = 0xF0 & ~(0xCC ^ 0xAA) & 0xFF;  // a & ~(b ^ c)
constexpr int TT_A_AND_B_NE_C = 0xF0 & (0xCC ^ 0xAA);          // a & (b ^ c)
constexpr int TT_A_AND_B_ANDN_C = 0xF0 & 0xCC & 0x55;          // a & b & ~c

// share = the two base sets intersect  <=>  NOT "certainly different" ((q & t) < 16 on bytes)
__device__ __forceinline__ uint32_t share_bits(const uint32_t *q, const uint32_t *t)
{
    return and_or(q[3], t[3], and_or(q[2], t[2], and_or(q[1], t[1], q[0] & t[0])));
}

struct MNHigh {  // src/measures.rs:14-23  d = #{(q&t) < 16}                       5 ops / 32 sites
    static constexpr int NP = 4, NC = 1, NT = 1, P0 = PL_A;  // planes A,G,C,T
    static __device__ __forceinline__ void step(uint32_t *a, const uint32_t *q, const uint32_t *t)
    {
        bcnt_acc(a[0], share_bits(q, t));
    }
    static __device__ __forceinline__ void tallies(const uint32_t *a, uint32_t total, uint32_t *o)
    {
        o[0] = total - a[0];
    }
};

struct MRaw {  // src/measures.rs:56-66  same: known & equal; else different      7 ops / 32 sites
    static constexpr int NP = 5, NC = 2, NT = 2, P0 = PL_A;  // planes A,G,C,T,K
    static __device__ __forceinline__ void step(uint32_t *a, const uint32_t *q, const uint32_t *t)
    {
        const uint32_t share = share_bits(q, t);
        bcnt_acc(a[0], share);
        bcnt_acc(a[1], bitop3<TT_AND3>(share, q[4], t[4]));  // both known and intersecting = equal
    }
    static __device__ __forceinline__ void tallies(const uint32_t *a, uint32_t total, uint32_t *o)
    {
        const uint32_t n = total - a[0];
        o[0] = n;         // n
        o[1] = n + a[1];  // d
    }
};

struct MK80 {  // src/measures.rs:85-107                                          7 ops / 32 sites
    static constexpr int NP = 4, NC = 3, NT = 3, P0 = PL_K;  // planes K,X1,X0,CL
    static __device__ __forceinline__ void step(uint32_t *a, const uint32_t *q, const uint32_t *t)
    {
        const uint32_t e1 = q[1] ^ t[1];                                 // classes differ
        const uint32_t sc = bitop3<TT_A_AND_B_ANDN_C>(q[0], t[0], e1);   // both known, same class
        const uint32_t ts = bitop3<TT_A_AND_B_NE_C>(sc, q[2], t[2]);     // A<->G or C<->T
        const uint32_t tv = bitop3<TT_AND3>(q[3], t[3], e1);             // purine vs pyrimidine class
        bcnt_acc(a[0], sc);
        bcnt_acc(a[1], ts);
        bcnt_acc(a[2], tv);
    }
    static __device__ __forceinline__ void tallies(const uint32_t *a, uint32_t, uint32_t *o)
    {
        o[0] = a[0] + a[2];  // count_L = same + ts + tv   (sc = same + ts)
        o[1] = a[1];         // ts
        o[2] = a[2];         // tv
    }
};

struct MTN93 {  // src/measures.rs:156-175                                        8 ops / 32 sites
    static constexpr int NP = 3, NC = 4, NT = 4, P0 = PL_K;  // planes K,X1,X0
    static __device__ __forceinline__ void step(uint32_t *a, const uint32_t *q, const uint32_t *t)
    {
        const uint32_t bk = q[0] & t[0];                                 // both known
        const uint32_t sc = bitop3<TT_A_AND_B_EQ_C>(bk, q[1], t[1]);     // same class
        const uint32_t ts = bitop3<TT_A_AND_B_NE_C>(sc, q[2], t[2]);     // transition
        const uint32_t p2 = ts & q[1];                                   // ... between pyrimidines
        bcnt_acc(a[0], bk);
        bcnt_acc(a[1], sc);
        bcnt_acc(a[2], ts);
        bcnt_acc(a[3], p2);
    }
    static __device__ __forceinline__ void tallies(const uint32_t *a, uint32_t, uint32_t *o)
    {
        o[0] = a[0];                  // count_L  (same + different, both known)
        o[1] = a[0] - (a[1] - a[2]);  // count_d  = L - same
        o[2] = a[2] - a[3];           // count_P1 (A<->G)
        o[3] = a[3];                  // count_P2 (C<->T)
    }
};

// One block = BM row records x 256*TN column records, 256 threads (4 waves), MINW blocks per CU
// (MINW = waves per SIMD asked of the register allocator).
// Per 128-site chunk:
//   columns: each lane loads its TN records' NP plane words straight from HBM/L2 into VGPRs
//            (16 B per lane, 1 KiB contiguous per wave-instruction) and keeps them for all BM rows;
//   rows:    the BM x NP uint4 row tile (<= 5 KiB) is staged through LDS, double-buffered, one
//            barrier per chunk; every lane reads the same 16 bytes (broadcast ds_read_b128), so
//            LDS time is 4 cycles per plane per row per wave against 64*TN VALU cycles;
//   tallies: BM*TN*NC accumulators live in VGPRs for the whole sweep over L.
//   epilogue: tallies (OUT_TALLY), int64 (OUT_INT) or — for the f64 measures — the finalisation
//            itself: the lane's tallies go through a lane-private LDS slot so that ONE copy of the
//            (log-heavy) finalisation code runs in a rolled loop instead of BM*TN inlined copies;
//            no tally round trip through HBM, no second kernel.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// 16 bytes per lane from (wave-uniform base) + (per-lane byte offset < 2^32): buffer_load_dwordx4 ... offen
__device__ __forceinline__ uint4 load_columns(const uint4 *uniform_base, uint32_t lane_byte)
{
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(uniform_base), 0, 0x7FFFFFFF, 0x00020000);
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_byte, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <class M, int BM, int TN, int MINW, int OUT>
__global__ __launch_bounds__(256, MINW) void pair_kernel(
    const uint4 *__restrict__ qpl, const uint4 *__restrict__ tpl,
    const BlockDesc *__restrict__ blocks, void *__restrict__ out_v,
    const uint32_t *__restrict__ q_counts, const uint32_t *__restrict__ t_counts, uint32_t nchunks,
    uint32_t q_npad, uint32_t t_npad, uint32_t n_cols, uint32_t row_begin, uint32_t row_end,
    uint64_t out_base, int square, uint32_t ksplit)
{
    constexpr int NP = M::NP, NC = M::NC, NT = M::NT;
    constexpr int QV = NP * BM;               // uint4 per staged row tile
    constexpr int QL = (QV + 255) / 256;      // staging loads per thread
    constexpr int RP = (BM % 2 == 0) ? BM / 2 : BM;   // rows per epilogue pass
    constexpr int STAGE_U4 = OUT >= 0 ? RP * TN * NT * 256 / 4 : 0;
    constexpr int SMEM_U4 = (2 * QV > STAGE_U4) ? 2 * QV : STAGE_U4;
    __shared__ uint4 smem[SMEM_U4];
    uint4 (*qs)[NP][BM] = reinterpret_cast<uint4 (*)[NP][BM]>(smem);

    // split-L launches (few tiles, long alignments): ksplit consecutive blocks share a tile and
    // each sweeps its own range of chunks; their partial tallies meet in integer atomics (exact).
    uint32_t tile = blockIdx.x, c_begin = 0, c_end = nchunks;
    if constexpr (OUT == OUT_TALLY_ADD || OUT == OUT_INT_ADD) {
        tile = blockIdx.x / ksplit;
        const uint32_t part = blockIdx.x - tile * ksplit;
        c_begin = (uint32_t)(((uint64_t)nchunks * part) / ksplit);
        c_end = (uint32_t)(((uint64_t)nchunks * (part + 1)) / ksplit);
    }
    const uint32_t i0 = __builtin_amdgcn_readfirstlane(blocks[tile].i0);
    const uint32_t j0 = __builtin_amdgcn_readfirstlane(blocks[tile].j0);
    if (i0 == 0xFFFFFFFFu || c_begin >= c_end)
        return;

    uint32_t acc[BM][TN][NC];
#pragma unroll
    for (int r = 0; r < BM; ++r)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int k = 0; k < NC; ++k)
                acc[r][tn][k] = 0;

    const size_t q_ps = (size_t)nchunks * q_npad;  // plane stride, in uint4
    const size_t t_ps = (size_t)nchunks * t_npad;
    // staging element(s) of this thread: (plane sp, row sr) of the row tile
    const uint4 *qsrc[QL];
    uint4 *qdst0[QL];
#pragma unroll
    for (int k = 0; k < QL; ++k) {
        const int e = (int)threadIdx.x + 256 * k;
        const int sp = e / BM, sr = e % BM;
        qsrc[k] = qpl + (size_t)(M::P0 + (e < QV ? sp : 0)) * q_ps + (size_t)c_begin * q_npad + i0 + sr;
        qdst0[k] = &qs[0][e < QV ? sp : 0][sr];
    }
    // column loads go through buffer descriptors: wave-uniform base in SGPRs (SALU arithmetic), the
    // lane's byte offset in ONE constant VGPR per column -> no VALU address arithmetic in the loop
    // (64-bit per-lane pointers cost two VALU adds per load, ~2 % of the issue slots)
    const uint4 *tchunk = tpl + (size_t)M::P0 * t_ps + (size_t)c_begin * t_npad + j0;
    uint32_t lane_byte[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
        lane_byte[tn] = (threadIdx.x + 256u * tn) * (uint32_t)sizeof(uint4);

    // prologue: first chunk of the row tile
#pragma unroll
    for (int k = 0; k < QL; ++k)
        if ((int)threadIdx.x + 256 * k < QV)
            *qdst0[k] = qsrc[k][0];
    __syncthreads();

#pragma unroll 1
    for (uint32_t c = 0; c < c_end - c_begin; ++c) {
        const uint32_t buf = c & 1u;
        const bool more = c + 1 < c_end - c_begin;
        uint4 qnext[QL];
        if (more) {
#pragma unroll
            for (int k = 0; k < QL; ++k)
                if ((int)threadIdx.x + 256 * k < QV)
                    qnext[k] = qsrc[k][(size_t)(c + 1) * q_npad];
        }
        uint4 tv[NP][TN];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                tv[p][tn] = load_columns(tchunk + (size_t)p * t_ps, lane_byte[tn]);
        tchunk += t_npad;

#pragma unroll
        for (int r = 0; r < BM; ++r) {
            uint4 qv[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p)
                qv[p] = qs[buf][p][r];  // same address in every lane: broadcast read
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                uint32_t q[NP], t[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) { q[p] = qv[p].x; t[p] = tv[p][tn].x; }
                M::step(acc[r][tn], q, t);
#pragma unroll
                for (int p = 0; p < NP; ++p) { q[p] = qv[p].y; t[p] = tv[p][tn].y; }
                M::step(acc[r][tn], q, t);
#pragma unroll
                for (int p = 0; p < NP; ++p) { q[p] = qv[p].z; t[p] = tv[p][tn].z; }
                M::step(acc[r][tn], q, t);
#pragma unroll
                for (int p = 0; p < NP; ++p) { q[p] = qv[p].w; t[p] = tv[p][tn].w; }
                M::step(acc[r][tn], q, t);
            }
        }
        if (more) {
#pragma unroll
            for (int k = 0; k < QL; ++k)
                if ((int)threadIdx.x + 256 * k < QV)
                    qdst0[k][(buf ^ 1u) * (NP * BM)] = qnext[k];
        }
        __syncthreads();
    }

    const uint32_t total = (c_end - c_begin) * kChunkSites;  // padded sites are N on both sides: "share"
    if constexpr (OUT < 0) {
#pragma unroll
        for (int r = 0; r < BM; ++r) {
            const uint32_t i = i0 + r;
            if (i >= row_end)
                break;
            const uint64_t row_at = square ? (tri_row_start(n_cols, i) - out_base) - (uint64_t)(i + 1)
                                           : (uint64_t)(i - row_begin) * n_cols;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const uint32_t j = j0 + 256 * tn + threadIdx.x;
                if (j < n_cols && (!square || j > i)) {
                    const uint64_t at = row_at + j;
                    uint32_t o[NT];
                    M::tallies(acc[r][tn], total, o);
                    if constexpr (OUT == OUT_INT) {
                        static_cast<int64_t *>(out_v)[at] = (int64_t)o[0];
                    } else if constexpr (OUT == OUT_INT_ADD) {
                        atomicAdd(static_cast<unsigned long long *>(out_v) + at, (unsigned long long)o[0]);
                    } else if constexpr (OUT == OUT_TALLY16) {
                        uint16_t *o16 = static_cast<uint16_t *>(out_v);
                        if constexpr (NT == 2) {
                            reinterpret_cast<uint32_t *>(o16)[at] = o[0] | (o[1] << 16);
                        } else if constexpr (NT == 4) {
                            reinterpret_cast<uint2 *>(o16)[at] = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
                        } else {
#pragma unroll
                            for (int k = 0; k < NT; ++k)
                                o16[at * NT + k] = (uint16_t)o[k];
                        }
                    } else if constexpr (OUT == OUT_TALLY_ADD) {
#pragma unroll
                        for (int k = 0; k < NT; ++k)
                            atomicAdd(static_cast<uint32_t *>(out_v) + at * NT + k, o[k]);
                    } else {
#pragma unroll
                        for (int k = 0; k < NT; ++k)
                            static_cast<uint32_t *>(out_v)[at * NT + k] = o[k];
                    }
                }
            }
        }
    } else {
        // every wave is past the loop's last barrier: the row-tile buffers are dead, reuse them.
        // Slot layout [pass-row][tn][k][lane]: a lane only ever touches its own column, so LDS
        // ordering within the wave is all the synchronisation needed.
        uint32_t *stage = reinterpret_cast<uint32_t *>(smem);
        double *out = static_cast<double *>(out_v);
#pragma unroll
        for (int pass = 0; pass < BM; pass += RP) {
#pragma unroll
            for (int rr = 0; rr < RP; ++rr)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    uint32_t o[NT];
                    M::tallies(acc[pass + rr][tn], total, o);
#pragma unroll
                    for (int k = 0; k < NT; ++k)
                        stage[((rr * TN + tn) * NT + k) * 256 + threadIdx.x] = o[k];
                }
#pragma unroll 1
            for (int e = 0; e < RP * TN; ++e) {
                const int rr = e / TN, tn = e % TN;
                const uint32_t i = i0 + pass + rr;
                if (i >= row_end)
                    break;
                const uint32_t j = j0 + 256 * tn + threadIdx.x;
                if (j < n_cols && (!square || j > i)) {
                    uint32_t o[NT];
#pragma unroll
                    for (int k = 0; k < NT; ++k)
                        o[k] = stage[(e * NT + k) * 256 + threadIdx.x];
                    uint4 qc = make_uint4(0, 0, 0, 0), tc = qc;
                    if constexpr (OUT == DST_TN93) {
                        qc = reinterpret_cast<const uint4 *>(q_counts)[i];
                        tc = reinterpret_cast<const uint4 *>(t_counts)[j];
                    }
                    const uint64_t at = square ? (tri_row_start(n_cols, i) - out_base) + (j - i - 1)
                                               : (uint64_t)(i - row_begin) * n_cols + j;
                    if constexpr (OUT == DST_RAW)
                        out[at] = finalize_pair<OUT>(o, qc, tc);
                    else   // out of line: the epilogue must not cost the sweep over L its registers
                        out[at] = finalize_pair_call<OUT>(o[0], o[1], NT > 2 ? o[2] : 0u, NT > 3 ? o[3] : 0u, qc, tc);
                }
            }
        }
    }
}

// tallies already in device memory -> the FloatInt payload (f64, or int64 for n / n_high): used
// after split-L launches (whose partial tallies meet in a scratch buffer) and by
// dst_finalize_device (tallies gathered from other GPUs).  One block per row of the range;
// threads stride along the row's pairs (coalesced).  T = uint32_t or uint16_t tallies.
template <int MEASURE, class T, bool CLOSE>
__global__ __launch_bounds__(256) void finalize_kernel(const T *__restrict__ tallies,
                                                       const uint32_t *__restrict__ q_counts,
                                                       const uint32_t *__restrict__ t_counts,
                                                       void *__restrict__ out_v, uint32_t n_cols,
                                                       uint32_t row_begin, uint64_t out_base,
                                                       int square)
{
    constexpr int NT = MEASURE == DST_N_HIGH ? 1 : MEASURE == DST_K80 ? 3 : MEASURE == DST_TN93 ? 4 : 2;
    const uint32_t i = row_begin + blockIdx.x;
    const uint32_t jstart = square ? i + 1 : 0;
    if (jstart >= n_cols)
        return;
    const uint64_t base = square ? tri_row_start(n_cols, i) - out_base
                                 : (uint64_t)(i - row_begin) * n_cols;
    uint4 qc = make_uint4(0, 0, 0, 0), tc = qc;
    if constexpr (MEASURE == DST_TN93)
        qc = reinterpret_cast<const uint4 *>(q_counts)[i];
    for (uint32_t k = threadIdx.x; k < n_cols - jstart; k += blockDim.x) {
        const uint64_t at = base + k;
        uint32_t o[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
            o[t] = tallies[at * NT + t];
        if constexpr (MEASURE == DST_N_HIGH) {
            static_cast<int64_t *>(out_v)[at] = (int64_t)o[0];
        } else {
            if constexpr (MEASURE == DST_TN93)
                tc = reinterpret_cast<const uint4 *>(t_counts)[jstart + k];
            static_cast<double *>(out_v)[at] = finalize_pair<MEASURE, CLOSE>(o, qc, tc);
        }
    }
}

}  // namespace

// =============================================================================================
// launchers
// =============================================================================================
hipError_t launch_pack(const uint8_t *d_codes, size_t row_stride, const DeviceSet &set,
                       unsigned long long *d_first_bad, const PackLists *lists, hipStream_t stream, size_t rec_begin, size_t rec_end)
{
    // the records of one rank's share; the rank that holds the set's last record also writes the padding records (all N)
    const uint32_t first = (uint32_t)std::min(rec_begin, set.npad);
    const uint32_t last = (uint32_t)(rec_end >= set.n ? set.npad : rec_end);
    if (last <= first)
        return hipSuccess;
    const PackLists none{};
    // bit 0: every row starts on a 16-byte boundary; bit 1: the matrix itself starts on a 4-byte boundary
    const int aligned16 = ((reinterpret_cast<uintptr_t>(d_codes) % 16 == 0) && (row_stride % 16 == 0) ? 1 : 0) |
                          (reinterpret_cast<uintptr_t>(d_codes) % 4 == 0 ? 2 : 0);
    dim3 grid((unsigned)set.nchunks, (unsigned)((last - first + 255) / 256));
    hipLaunchKernelGGL(pack_kernel, grid, dim3(256), 0, stream, d_codes, row_stride, (uint32_t)set.n,
                       (uint32_t)set.len, (uint32_t)set.nchunks, (uint32_t)set.npad, set.planes,
                       d_first_bad, aligned16, lists ? *lists : none, first, last);
    return hipGetLastError();
}

hipError_t launch_pack_nibbles(const uint8_t *d_nibbles, size_t row_stride, const DeviceSet &set, unsigned long long *d_first_bad,
                               hipStream_t stream)
{
    dim3 grid((unsigned)set.nchunks, (unsigned)((set.npad + 255) / 256));
    hipLaunchKernelGGL(pack_nibbles_kernel, grid, dim3(256), 0, stream, d_nibbles, row_stride, (uint32_t)set.n, (uint32_t)set.len,
                       (uint32_t)set.nchunks, (uint32_t)set.npad, set.planes, d_first_bad);
    return hipGetLastError();
}

hipError_t launch_derive(const DeviceSet &set, hipStream_t stream)
{
    const size_t per_plane = set.nchunks * set.npad;
    hipLaunchKernelGGL(derive_kernel, dim3((unsigned)((per_plane + 255) / 256)), dim3(256), 0, stream, set.planes, per_plane);
    return hipGetLastError();
}

hipError_t launch_planes_from_slots(const DeviceSet &set, hipStream_t stream)
{
    dim3 grid((unsigned)set.nchunks, (unsigned)((set.npad + 255) / 256));
    hipLaunchKernelGGL(planes_from_slots_kernel, grid, dim3(256), 0, stream, set.rec.pre_slots, set.ref.planes, (uint32_t)set.n,
                       (uint32_t)set.nchunks, (uint32_t)set.npad, set.planes);
    return hipGetLastError();
}

hipError_t launch_range_counts(const DeviceSet &set, size_t rec_begin, size_t rec_end, uint32_t *out, hipStream_t stream,
                               const PackLists *lists)
{
    // (records past the set's own, the padding up to npad, count nothing)
    if (rec_end <= rec_begin)
        return hipSuccess;
    const bool slots = lists && lists->defer_planes;
    hipLaunchKernelGGL(counts_kernel, dim3((unsigned)((rec_end - rec_begin + 31) / 32)), dim3(256), 0, stream, set.planes,
                       slots ? lists->slots : nullptr, slots ? lists->ref_planes : nullptr, slots ? lists->stats : nullptr,
                       slots ? lists->max_dev_sum : 0ull, (uint32_t)set.n, (uint32_t)set.nchunks, (uint32_t)set.npad, out,
                       (uint32_t)rec_begin, (uint32_t)rec_end);
    return hipGetLastError();
}

hipError_t launch_fill_counts(const DeviceSet &set, hipStream_t stream, const PackLists *lists)
{
    return launch_range_counts(set, 0, set.npad, set.counts, stream, lists);
}

namespace {

struct Variant {
    int bm, tn, minw;  // rows per tile, columns per lane, waves per SIMD asked of the register allocator
};
// [0] is the default: picked on MI355X with tools/kbench.py (profiles/r01/kbench_variants.txt)
constexpr Variant kVarNHigh[] = {{24, 2, 4}, {16, 2, 4}, {32, 2, 3}};
constexpr Variant kVarRaw[] = {{12, 2, 4}, {16, 2, 3}, {32, 2, 2}};
constexpr Variant kVarK80[] = {{8, 2, 3}, {12, 2, 3}};
// (r03: the out-of-line finalisation costs the 8-row tile its fourth wave per SIMD: 261 ms at 50,000 x 30,000 against 249
// for 12 rows at three waves, which is the default now)
constexpr Variant kVarTN93[] = {{12, 2, 3}, {8, 2, 4}, {16, 2, 2}};

template <class M, int BM, int TN, int MINW, int OUT>
hipError_t launch_one(const PairLaunch &pl, hipStream_t stream)
{
    hipLaunchKernelGGL((pair_kernel<M, BM, TN, MINW, OUT>), dim3(pl.nblocks), dim3(256), 0, stream,
                       pl.rows->planes, pl.cols->planes, pl.d_blocks, pl.d_out, pl.rows->counts,
                       pl.cols->counts, (uint32_t)pl.rows->nchunks, (uint32_t)pl.rows->npad,
                       (uint32_t)pl.cols->npad, (uint32_t)pl.cols->n, (uint32_t)pl.row_begin,
                       (uint32_t)pl.row_end, pl.out_base, pl.square ? 1 : 0, (uint32_t)pl.ksplit);
    return hipGetLastError();
}

template <class M, int BM, int TN, int MINW, int OUT>
hipError_t launch_split(const PairLaunch &pl, hipStream_t stream)
{
    hipLaunchKernelGGL((pair_kernel<M, BM, TN, MINW, OUT>), dim3(pl.nblocks * (unsigned)pl.ksplit), dim3(256), 0,
                       stream, pl.rows->planes, pl.cols->planes, pl.d_blocks, pl.d_out, pl.rows->counts,
                       pl.cols->counts, (uint32_t)pl.rows->nchunks, (uint32_t)pl.rows->npad,
                       (uint32_t)pl.cols->npad, (uint32_t)pl.cols->n, (uint32_t)pl.row_begin,
                       (uint32_t)pl.row_end, pl.out_base, pl.square ? 1 : 0, (uint32_t)pl.ksplit);
    return hipGetLastError();
}

// every output form of one (family, tile variant)
template <class M, int BM, int TN, int MINW>
hipError_t launch_outputs(int measure, const PairLaunch &pl, hipStream_t stream)
{
    if (pl.ksplit > 1) {  // the caller zeroed the buffer; f64 goes through a tally scratch
        if constexpr (M::NT == 1) {
            if (pl.out_kind == DST_OUT_DISTANCE)
                return launch_split<M, BM, TN, MINW, OUT_INT_ADD>(pl, stream);
        }
        return launch_split<M, BM, TN, MINW, OUT_TALLY_ADD>(pl, stream);
    }
    if (pl.out_kind == DST_OUT_TALLY)
        return launch_one<M, BM, TN, MINW, OUT_TALLY>(pl, stream);
    if (pl.out_kind == DST_OUT_TALLY16)
        return launch_one<M, BM, TN, MINW, OUT_TALLY16>(pl, stream);
    if constexpr (M::NT == 1) {
        return launch_one<M, BM, TN, MINW, OUT_INT>(pl, stream);
    } else if constexpr (M::NT == 2) {
        return measure == DST_RAW ? launch_one<M, BM, TN, MINW, DST_RAW>(pl, stream)
                                  : launch_one<M, BM, TN, MINW, DST_JC69>(pl, stream);
    } else if constexpr (M::NT == 3) {
        return launch_one<M, BM, TN, MINW, DST_K80>(pl, stream);
    } else {
        return launch_one<M, BM, TN, MINW, DST_TN93>(pl, stream);
    }
}

}  // namespace

int variant_count(int measure)
{
    switch (measure) {
    case DST_N:
    case DST_N_HIGH: return (int)(sizeof kVarNHigh / sizeof kVarNHigh[0]);
    case DST_RAW:
    case DST_JC69: return (int)(sizeof kVarRaw / sizeof kVarRaw[0]);
    case DST_K80: return (int)(sizeof kVarK80 / sizeof kVarK80[0]);
    case DST_TN93: return (int)(sizeof kVarTN93 / sizeof kVarTN93[0]);
    default: return 0;
    }
}

static Variant pick_variant(int measure, int variant)
{
    const int nv = variant_count(measure);
    if (variant < 0 || variant >= nv)
        variant = 0;
    switch (measure) {
    case DST_N:
    case DST_N_HIGH: return kVarNHigh[variant];
    case DST_RAW:
    case DST_JC69: return kVarRaw[variant];
    case DST_K80: return kVarK80[variant];
    default: return kVarTN93[variant];
    }
}

TileShape tile_shape(int measure, int variant)
{
    const Variant v = pick_variant(measure, variant);
    return TileShape{v.bm, 256 * v.tn};
}

#define DST_CASE(M, BM_, TN_, W_)                                   \
    if (v.bm == BM_ && v.tn == TN_ && v.minw == W_)                 \
        return launch_outputs<M, BM_, TN_, W_>(measure, pl, stream);

hipError_t launch_pairs(int measure, int variant, const PairLaunch &pl, hipStream_t stream)
{
    const Variant v = pick_variant(measure, variant);
    switch (measure) {
    case DST_N:
    case DST_N_HIGH:
        DST_CASE(MNHigh, 24, 2, 4) DST_CASE(MNHigh, 16, 2, 4) DST_CASE(MNHigh, 32, 2, 3)
        break;
    case DST_RAW:
    case DST_JC69:
        DST_CASE(MRaw, 12, 2, 4) DST_CASE(MRaw, 16, 2, 3) DST_CASE(MRaw, 32, 2, 2)
        break;
    case DST_K80:
        DST_CASE(MK80, 12, 2, 3) DST_CASE(MK80, 8, 2, 3)
        break;
    case DST_TN93:
        DST_CASE(MTN93, 12, 2, 3) DST_CASE(MTN93, 8, 2, 4) DST_CASE(MTN93, 16, 2, 2)
        break;
    default: break;
    }
    return hipErrorInvalidValue;
}
#undef DST_CASE

hipError_t launch_finalize(int measure, const PairLaunch &pl, const void *d_tallies, bool tallies16,
                           void *d_out, hipStream_t stream, bool close)
{
    const unsigned rows = (unsigned)(pl.row_end - pl.row_begin);
    if (rows == 0)
        return hipSuccess;
#define DST_FIN2(MEAS, T, CL)                                                                                        \
    hipLaunchKernelGGL((finalize_kernel<MEAS, T, CL>), dim3(rows), dim3(256), 0, stream, static_cast<const T *>(d_tallies), \
                       pl.rows->counts, pl.cols->counts, d_out, (uint32_t)pl.cols->n, (uint32_t)pl.row_begin, pl.out_base, \
                       pl.square ? 1 : 0)
#define DST_FIN(MEAS)                             \
    do {                                          \
        if (tallies16 && close)                   \
            DST_FIN2(MEAS, uint16_t, true);       \
        else if (tallies16)                       \
            DST_FIN2(MEAS, uint16_t, false);      \
        else if (close)                           \
            DST_FIN2(MEAS, uint32_t, true);       \
        else                                      \
            DST_FIN2(MEAS, uint32_t, false);      \
    } while (0)
    switch (measure) {
    case DST_N:
    case DST_N_HIGH: DST_FIN(DST_N_HIGH); break;
    case DST_RAW: DST_FIN(DST_RAW); break;
    case DST_JC69: DST_FIN(DST_JC69); break;
    case DST_K80: DST_FIN(DST_K80); break;
    case DST_TN93: DST_FIN(DST_TN93); break;
    default: return hipErrorInvalidValue;
    }
#undef DST_FIN
#undef DST_FIN2
    return hipGetLastError();
}

}  // namespace dst
