// dst_consensus.hip — the consensus-delta path: hand-written gfx950 kernels that compute the same
// integer tallies as the dense bit-plane kernels (dst_kernels.hip) from each record's DIFFERENCES to a
// reference sequence instead of from all L sites.
//
// The reference's own fastest mode does this for one measure: `-m n` walks the two records' lists of
// differences from the alignment's consensus (src/measures.rs:28-53, lists from get_differences(),
// src/fastaio.rs:67-75, consensus from src/fastaio.rs:289-336).  Here the idea carries every measure:
// each tally is a sum over sites of a per-site function f_k of the two codes' high nibbles, so against
// ANY reference sequence c
//
//     T_k(q,t) = F_k + A_k(q) + A_k(t) + sum over the sites where BOTH q and t differ from c of h_k
//
// (dst_internal.h has the terms).  F and A are per-record constants; only the intersection term is
// pairwise, and for low-diversity alignments (SARS-CoV-2-like: ~65 differing sites of 30,000 per record)
// it touches a few columns per row.  The pair kernel is then bound by WRITING the N^2/2 results:
//
//   ref_sample_kernel   per-site plurality code over a sample of records -> the reference c
//   index_kernel        per record: ascending list of (site, nibble) where it differs from c; and the
//                       same entries bucketed by (site, panel of 8,192 column records)
//   scan kernels        exclusive scan of the list lengths -> CSR offsets
//   aconst_kernel       A_k(record), packed like the accumulators
//   consensus_pair_kernel  one block = a few rows x one column panel: the row's list is joined with the
//                       site buckets of the panel, h_k goes into LDS accumulators (ds_add_u32), then one
//                       coalesced pass adds the per-record constants, finalises (f64, reference operation
//                       order) and stores in canonical order
//   site_hist_kernel    exact per-site base counts for consensus() itself (dst_consensus)
//
// Integer adds only (order-independent, exact); two 16-bit tallies share one 32-bit accumulator when the
// alignment is shorter than 65,536 sites (the sums are exact modulo 2^32 and every final tally fits).
#include "dst_device.hpp"

namespace dst {
namespace {

__device__ __forceinline__ int ref_class(uint32_t nib)  // A G C T N-class -> 0..4
{
    return nib == 8 ? 0 : nib == 4 ? 1 : nib == 2 ? 2 : nib == 1 ? 3 : 4;
}

__device__ __forceinline__ uint32_t popc4(uint4 v)
{
    return __builtin_popcount(v.x) + __builtin_popcount(v.y) + __builtin_popcount(v.z) + __builtin_popcount(v.w);
}

// =============================================================================================
// reference sequence: per-site plurality over a sample of the records
// =============================================================================================
// One block = one 128-site chunk, one thread = one site.  Every thread of a block reads the same 16
// bytes of a sampled record (broadcast).  Classes: known A, G, C, T and the N class (N, -, ?); ties go
// to the first in that order.  Any choice gives exact results — the reference only decides how much
// work the pair kernel has.
__global__ __launch_bounds__(128) void ref_sample_kernel(const uint32_t *__restrict__ planes32, uint32_t n,
                                                         uint32_t len, uint32_t nchunks, uint32_t npad,
                                                         uint32_t samples, uint8_t *__restrict__ ref_nib,
                                                         uint4 *__restrict__ ref_planes,
                                                         unsigned long long *__restrict__ stats)
{
    const uint32_t c = blockIdx.x, b = threadIdx.x;
    const uint32_t w = b >> 5, bit = b & 31;
    const size_t ps = (size_t)nchunks * npad * 4;  // plane stride in 32-bit words
    uint32_t cnt[5] = {0, 0, 0, 0, 0};
    for (uint32_t k = 0; k < samples; ++k) {
        const uint32_t r = (uint32_t)(((uint64_t)k * n) / samples);
        const size_t at = ((size_t)c * npad + r) * 4 + w;
        const uint32_t A = (planes32[PL_A * ps + at] >> bit) & 1u, G = (planes32[PL_G * ps + at] >> bit) & 1u;
        const uint32_t C = (planes32[PL_C * ps + at] >> bit) & 1u, T = (planes32[PL_T * ps + at] >> bit) & 1u;
        const uint32_t nib = A << 3 | G << 2 | C << 1 | T;
        cnt[0] += nib == 8;
        cnt[1] += nib == 4;
        cnt[2] += nib == 2;
        cnt[3] += nib == 1;
        cnt[4] += nib == 15;
    }
    uint32_t best = cnt[0], cls = 0;
#pragma unroll
    for (uint32_t k = 1; k < 5; ++k)
        if (cnt[k] > best) {
            best = cnt[k];
            cls = k;
        }
    const uint32_t site = c * kChunkSites + b;
    const bool real = site < len;
    const uint32_t nib = !real ? 15u : cls == 0 ? 8u : cls == 1 ? 4u : cls == 2 ? 2u : cls == 3 ? 1u : 15u;
    ref_nib[site] = (uint8_t)nib;
    const uint32_t wave = b >> 6;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned long long m = __ballot((nib >> (3 - p)) & 1u);
        if ((b & 63) == 0)
            reinterpret_cast<unsigned long long *>(&ref_planes[(size_t)p * nchunks + c])[wave] = m;
    }
    // statistics for the path choice: known reference sites, sum and sum of squares of the sampled
    // records that deviate from the plurality class
    const unsigned long long known = __ballot(real && cls < 4);
    uint32_t dev = real ? samples - best : 0u, dev2 = dev * dev;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dev += __shfl_xor(dev, o);
        dev2 += __shfl_xor(dev2, o);
    }
    if ((b & 63) == 0) {
        atomicAdd(&stats[0], (unsigned long long)__builtin_popcountll(known));
        atomicAdd(&stats[1], (unsigned long long)dev);
        atomicAdd(&stats[2], (unsigned long long)dev2);
    }
    if (c == 0 && b == 0)
        stats[3] = samples;
}

// =============================================================================================
// difference lists
// =============================================================================================
// One wave = 8 records x 8 chunks per step (lane = chunk-lane * 8 + record-lane), so the eight lanes of
// a chunk read one 128-byte line of each plane, and a record's entries come out in ascending site order:
// the exclusive prefix over the chunk-lanes of a record is three shuffles.
// FILL == false: rec[r] = list length, site[b] += 1 per entry (b = site * n_panels + r / kPanelCols).
// FILL == true : rec = scanned offsets, site = scanned bucket offsets, site_cur = zeroed cursors.
template <bool FILL>
__global__ __launch_bounds__(256) void index_kernel(const uint4 *__restrict__ planes,
                                                    const uint4 *__restrict__ ref_planes, uint32_t n,
                                                    uint32_t nchunks, uint32_t npad, int want_sites,
                                                    int skip_nclass, uint32_t *__restrict__ rec,
                                                    uint32_t *__restrict__ rec_ent, uint32_t *__restrict__ site,
                                                    uint32_t *__restrict__ site_cur,
                                                    uint32_t *__restrict__ site_ent, uint32_t n_panels,
                                                    unsigned long long *__restrict__ total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t rl = lane & 7u, cl = lane >> 3;
    const uint32_t r = wave * 8u + rl;
    const bool live = r < n;
    const size_t ps = (size_t)nchunks * npad;
    const uint32_t panel = r / kPanelCols;
    uint32_t run = 0;
    const uint32_t base0 = (FILL && live) ? rec[r] : 0u;
    for (uint32_t c0 = 0; c0 < nchunks; c0 += 8) {
        const uint32_t c = c0 + cl;
        uint4 A = make_uint4(0, 0, 0, 0), G = A, C = A, T = A, d = A;
        if (live && c < nchunks) {
            const size_t at = (size_t)c * npad + r;
            A = planes[PL_A * ps + at];
            G = planes[PL_G * ps + at];
            C = planes[PL_C * ps + at];
            T = planes[PL_T * ps + at];
            const uint4 rA = ref_planes[c], rG = ref_planes[nchunks + c];
            const uint4 rC = ref_planes[2 * (size_t)nchunks + c], rT = ref_planes[3 * (size_t)nchunks + c];
            d.x = (A.x ^ rA.x) | (G.x ^ rG.x) | (C.x ^ rC.x) | (T.x ^ rT.x);
            d.y = (A.y ^ rA.y) | (G.y ^ rG.y) | (C.y ^ rC.y) | (T.y ^ rT.y);
            d.z = (A.z ^ rA.z) | (G.z ^ rG.z) | (C.z ^ rC.z) | (T.z ^ rT.z);
            d.w = (A.w ^ rA.w) | (G.w ^ rG.w) | (C.w ^ rC.w) | (T.w ^ rT.w);
            if (skip_nclass) {  // get_differences(): seq[i] < 240 (src/fastaio.rs:70)
                d.x &= ~(A.x & G.x & C.x & T.x);
                d.y &= ~(A.y & G.y & C.y & T.y);
                d.z &= ~(A.z & G.z & C.z & T.z);
                d.w &= ~(A.w & G.w & C.w & T.w);
            }
        }
        const uint32_t pc = popc4(d);
        uint32_t incl = pc, up;
        up = __shfl_up(incl, 8);
        if (cl >= 1) incl += up;
        up = __shfl_up(incl, 16);
        if (cl >= 2) incl += up;
        up = __shfl_up(incl, 32);
        if (cl >= 4) incl += up;
        const uint32_t tot = __shfl(incl, 56 + rl);
        if (pc) {
            uint32_t at = base0 + run + (incl - pc);
            const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
            const uint32_t aw[4] = {A.x, A.y, A.z, A.w}, gw[4] = {G.x, G.y, G.z, G.w};
            const uint32_t cw[4] = {C.x, C.y, C.z, C.w}, tw[4] = {T.x, T.y, T.z, T.w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                uint32_t m = dw[w];
                while (m) {
                    const uint32_t bit = (uint32_t)__builtin_ctz(m);
                    m &= m - 1;
                    const uint32_t s = c * kChunkSites + 32u * w + bit;
                    if constexpr (FILL) {
                        const uint32_t nib = ((aw[w] >> bit) & 1u) << 3 | ((gw[w] >> bit) & 1u) << 2 |
                                             ((cw[w] >> bit) & 1u) << 1 | ((tw[w] >> bit) & 1u);
                        rec_ent[at++] = s | nib << kEntryShift;
                        if (want_sites) {
                            const size_t bk = (size_t)s * n_panels + panel;
                            const uint32_t pos = atomicAdd(&site_cur[bk], 1u);
                            site_ent[site[bk] + pos] = r | nib << kEntryShift;
                        }
                    } else if (want_sites) {
                        atomicAdd(&site[(size_t)s * n_panels + panel], 1u);
                    }
                }
            }
        }
        run += tot;
    }
    if constexpr (!FILL) {
        if (live && cl == 0)
            rec[r] = run;
        // the 8 record-lanes of chunk-lane 0 hold the 8 list lengths of this wave
        uint32_t sum = (live && cl == 0) ? run : 0u;
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        sum += __shfl_xor(sum, 4);
        if (lane == 0 && sum)
            atomicAdd(total, (unsigned long long)sum);
    }
}

// =============================================================================================
// exclusive scan (list lengths -> offsets)
// =============================================================================================
constexpr uint32_t kScanPerBlock = 2048;  // 256 threads x 8

__global__ __launch_bounds__(256) void scan_block_kernel(uint32_t *__restrict__ data, size_t n,
                                                         uint32_t *__restrict__ block_sums)
{
    __shared__ uint32_t wave_tot[4];
    const size_t base = (size_t)blockIdx.x * kScanPerBlock + (size_t)threadIdx.x * 8;
    uint32_t v[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        v[k] = base + k < n ? data[base + k] : 0u;
        sum += v[k];
    }
    uint32_t incl = sum, up;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        up = __shfl_up(incl, o);
        if ((threadIdx.x & 63u) >= (uint32_t)o) incl += up;
    }
    if ((threadIdx.x & 63u) == 63u)
        wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t off = 0;
    for (uint32_t wv = 0; wv < (threadIdx.x >> 6); ++wv)
        off += wave_tot[wv];
    uint32_t run = off + incl - sum;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (base + k < n)
            data[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 255)
        block_sums[blockIdx.x] = off + incl;
}

__global__ __launch_bounds__(256) void scan_add_kernel(uint32_t *__restrict__ data, size_t n,
                                                       const uint32_t *__restrict__ block_offsets)
{
    const size_t base = (size_t)blockIdx.x * kScanPerBlock + (size_t)threadIdx.x * 8;
    const uint32_t off = block_offsets[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (base + k < n)
            data[base + k] += off;
}

// =============================================================================================
// per-record constants A_k
// =============================================================================================
__global__ __launch_bounds__(256) void aconst_kernel(const uint32_t *__restrict__ off,
                                                     const uint32_t *__restrict__ ent,
                                                     const uint8_t *__restrict__ ref_nib,
                                                     const ConsensusLut *__restrict__ lut, int family, int wide,
                                                     int words, uint32_t n, uint32_t npad,
                                                     uint32_t *__restrict__ aconst)
{
    const uint32_t lane = threadIdx.x & 63u, r = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (r >= n)
        return;
    uint32_t acc[kMaxWords] = {0, 0, 0, 0};
    for (uint32_t i = off[r] + lane; i < off[r + 1]; i += 64) {
        const uint32_t e = ent[i];
        const uint32_t *a = lut->a[family][wide][ref_class(ref_nib[e & kEntryMask])][e >> kEntryShift];
#pragma unroll
        for (int w = 0; w < kMaxWords; ++w)
            acc[w] += a[w];
    }
#pragma unroll
    for (int w = 0; w < kMaxWords; ++w) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            acc[w] += __shfl_xor(acc[w], o);
        if (lane == 0 && w < words)
            aconst[(size_t)w * npad + r] = acc[w];
    }
}

// =============================================================================================
// pair kernel
// =============================================================================================
template <int FAM, bool WIDE>
struct Pack {
    static constexpr int NT = FAM == FAM_NHIGH ? 1 : FAM == FAM_RAW ? 2 : FAM == FAM_K80 ? 3 : 4;
    static constexpr int W = WIDE ? NT : (NT + 1) / 2;
    static __device__ __forceinline__ void unpack(const uint32_t *t, uint32_t *o)
    {
        if constexpr (WIDE || NT == 1) {
#pragma unroll
            for (int k = 0; k < NT; ++k)
                o[k] = t[k];
        } else {
            o[0] = t[0] & 0xFFFFu;
            o[1] = t[0] >> 16;
            if constexpr (NT == 3) {
                o[2] = t[1];
            } else if constexpr (NT == 4) {
                o[2] = t[1] & 0xFFFFu;
                o[3] = t[1] >> 16;
            }
        }
    }
};

struct FWords {
    uint32_t w[kMaxWords];
};

// One block = rows [i0, i1) x one panel of up to kPanelCols column records, 256 threads.
// Per row:  A) each thread takes one entry (site, nibble) of the row's list and looks up the panel's
//              bucket of that site; a block-wide exclusive scan of the bucket sizes numbers the candidate
//              events;
//           B) the events are dealt to the threads (binary search in the scanned sizes): column record
//              and nibble from the bucket, h_k from the table, ds_add_u32 into the column's accumulators;
//           C) every thread walks its columns (coalesced): accumulator + A(column) + A(row) + F, unpack,
//              finalise, store in canonical order; touched accumulators are reset on the way.
// A row's list longer than 256 entries repeats A/B in slices.
template <int FAM, bool WIDE, int OUT>
__global__ __launch_bounds__(256) void consensus_pair_kernel(
    const uint32_t *__restrict__ row_off, const uint32_t *__restrict__ row_ent,
    const uint32_t *__restrict__ row_a, uint32_t row_npad, const uint32_t *__restrict__ site_off,
    const uint32_t *__restrict__ site_ent, const uint32_t *__restrict__ col_a, uint32_t col_npad,
    uint32_t n_panels, const uint8_t *__restrict__ ref_nib, const ConsensusLut *__restrict__ lut, FWords fw,
    const ConsensusTile *__restrict__ tiles, void *__restrict__ out_v, const uint32_t *__restrict__ q_counts,
    const uint32_t *__restrict__ t_counts, uint32_t n_cols, uint32_t row_begin, uint64_t out_base, int square)
{
    using P = Pack<FAM, WIDE>;
    constexpr int W = P::W, NT = P::NT;
    extern __shared__ uint32_t smem[];
    uint32_t *acc = smem;                        // [W][kPanelCols]
    uint32_t *seg_start = smem + W * kPanelCols;  // [257] scanned bucket sizes of the current slice
    uint32_t *seg_o0 = seg_start + 257;          // [256] bucket start in site_ent
    uint32_t *seg_meta = seg_o0 + 256;           // [256] reference class * 16 + row nibble
    uint32_t *wave_tot = seg_meta + 256;         // [4]

    const ConsensusTile tile = tiles[blockIdx.x];
    const uint32_t panel0 = tile.panel * kPanelCols;
    const uint32_t pcols = min(kPanelCols, n_cols - panel0);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t k = tid; k < W * kPanelCols; k += 256)
        acc[k] = 0;
    __syncthreads();

    for (uint32_t q = tile.i0; q < tile.i1; ++q) {
        const uint32_t e0 = row_off[q], e1 = row_off[q + 1];
        for (uint32_t eb = e0; eb < e1; eb += 256) {
            // ---- A: bucket of every entry of the slice, scanned
            uint32_t cnt = 0, o0 = 0, meta = 0;
            if (eb + tid < e1) {
                const uint32_t e = row_ent[eb + tid];
                const uint32_t s = e & kEntryMask;
                const size_t bk = (size_t)s * n_panels + tile.panel;
                o0 = site_off[bk];
                cnt = site_off[bk + 1] - o0;
                meta = (uint32_t)ref_class(ref_nib[s]) * 16u + (e >> kEntryShift);
            }
            uint32_t incl = cnt, up;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                up = __shfl_up(incl, o);
                if (lane >= (uint32_t)o) incl += up;
            }
            if (lane == 63)
                wave_tot[wave] = incl;
            __syncthreads();
            uint32_t woff = 0, total = 0;
#pragma unroll
            for (uint32_t wv = 0; wv < 4; ++wv) {
                const uint32_t t = wave_tot[wv];
                if (wv < wave) woff += t;
                total += t;
            }
            seg_start[tid] = woff + incl - cnt;
            seg_o0[tid] = o0;
            seg_meta[tid] = meta;
            if (tid == 0)
                seg_start[256] = total;
            __syncthreads();
            // ---- B: one candidate event per thread per round
            for (uint32_t i = tid; i < total; i += 256) {
                uint32_t lo = 0, hi = 256;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (seg_start[mid] <= i) lo = mid; else hi = mid;
                }
                const uint32_t ce = site_ent[seg_o0[lo] + (i - seg_start[lo])];
                const uint32_t t = ce & kEntryMask;
                if (!square || t > q) {
                    const uint32_t m = seg_meta[lo];
                    const uint32_t *h = lut->h[FAM][WIDE ? 1 : 0][m >> 4][m & 15u][ce >> kEntryShift];
#pragma unroll
                    for (int w = 0; w < W; ++w) {
                        const uint32_t v = h[w];
                        if (v)
                            atomicAdd(&acc[w * kPanelCols + (t - panel0)], v);
                    }
                }
            }
            __syncthreads();
        }
        // ---- C: constants, finalisation, canonical-order store
        uint32_t aq[W];
#pragma unroll
        for (int w = 0; w < W; ++w)
            aq[w] = row_a[(size_t)w * row_npad + q] + fw.w[w];
        uint4 qc = make_uint4(0, 0, 0, 0);
        if constexpr (OUT == DST_TN93)
            qc = reinterpret_cast<const uint4 *>(q_counts)[q];
        const uint64_t row_at = square ? (tri_row_start(n_cols, q) - out_base) - (uint64_t)(q + 1)
                                       : (uint64_t)(q - row_begin) * n_cols;
        // square: the row's first live column of this panel is q + 1
        uint32_t k0 = tid;
        if (square && q + 1 > panel0) {
            const uint32_t skip = q + 1 - panel0;           // columns [0, skip) of the panel pair with nothing
            k0 = (skip & ~255u) + tid;
            if (k0 < skip) k0 += 256;
        }
        for (uint32_t k = k0; k < pcols; k += 256) {
            const uint32_t t = panel0 + k;
            uint32_t tot[W], o[NT];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const uint32_t a = acc[w * kPanelCols + k];
                if (a)
                    acc[w * kPanelCols + k] = 0;
                tot[w] = a + col_a[(size_t)w * col_npad + t] + aq[w];
            }
            P::unpack(tot, o);
            const uint64_t at = row_at + t;
            if constexpr (OUT == OUT_INT) {
                static_cast<int64_t *>(out_v)[at] = (int64_t)o[0];
            } else if constexpr (OUT == OUT_TALLY) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    static_cast<uint32_t *>(out_v)[at * NT + j] = o[j];
            } else if constexpr (OUT == OUT_TALLY16) {
                uint16_t *o16 = static_cast<uint16_t *>(out_v);
                if constexpr (NT == 2) {
                    reinterpret_cast<uint32_t *>(o16)[at] = o[0] | (o[1] << 16);
                } else if constexpr (NT == 4) {
                    reinterpret_cast<uint2 *>(o16)[at] = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
                } else {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        o16[at * NT + j] = (uint16_t)o[j];
                }
            } else {
                uint4 tc = make_uint4(0, 0, 0, 0);
                if constexpr (OUT == DST_TN93)
                    tc = reinterpret_cast<const uint4 *>(t_counts)[t];
                static_cast<double *>(out_v)[at] = finalize_pair<OUT>(o, qc, tc);
            }
        }
        __syncthreads();
    }
}

// =============================================================================================
// exact per-site counts for consensus() (src/fastaio.rs:289-336)
// =============================================================================================
// hist[site][3] += records of this block's range whose code at the site is G, C, T (72, 40, 24); every
// other byte goes to the A bucket (`lookup` maps it to 0), i.e. A = records - G - C - T.
constexpr uint32_t kHistRecords = 2048;
__global__ __launch_bounds__(128) void site_hist_kernel(const uint32_t *__restrict__ planes32, uint32_t n,
                                                        uint32_t len, uint32_t nchunks, uint32_t npad,
                                                        uint32_t *__restrict__ hist)
{
    const uint32_t c = blockIdx.x % nchunks, b = threadIdx.x, w = b >> 5, bit = b & 31;
    const uint32_t site = c * kChunkSites + b;
    const size_t ps = (size_t)nchunks * npad * 4;
    const uint32_t r0 = (blockIdx.x / nchunks) * kHistRecords, r1 = min(n, r0 + kHistRecords);
    uint32_t g = 0, cc = 0, t = 0;
    for (uint32_t r = r0; r < r1; ++r) {
        const size_t at = ((size_t)c * npad + r) * 4 + w;
        const uint32_t A = (planes32[PL_A * ps + at] >> bit) & 1u, G = (planes32[PL_G * ps + at] >> bit) & 1u;
        const uint32_t C = (planes32[PL_C * ps + at] >> bit) & 1u, T = (planes32[PL_T * ps + at] >> bit) & 1u;
        const uint32_t nib = A << 3 | G << 2 | C << 1 | T;
        g += nib == 4;
        cc += nib == 2;
        t += nib == 1;
    }
    if (site < len) {
        if (g) atomicAdd(&hist[(size_t)site * 3 + 0], g);
        if (cc) atomicAdd(&hist[(size_t)site * 3 + 1], cc);
        if (t) atomicAdd(&hist[(size_t)site * 3 + 2], t);
    }
}

}  // namespace

// =============================================================================================
// launchers
// =============================================================================================
hipError_t launch_ref_sample(const DeviceSet &set, hipStream_t stream)
{
    const uint32_t samples = (uint32_t)std::min<size_t>(set.n, 512);
    hipLaunchKernelGGL(ref_sample_kernel, dim3((unsigned)set.nchunks), dim3(128), 0, stream,
                       reinterpret_cast<const uint32_t *>(set.planes), (uint32_t)set.n, (uint32_t)set.len,
                       (uint32_t)set.nchunks, (uint32_t)set.npad, samples, set.ref.nib, set.ref.planes,
                       reinterpret_cast<unsigned long long *>(set.ref.stats));
    return hipGetLastError();
}

hipError_t launch_index(const DeviceSet &set, const uint4 *ref_planes, bool fill, bool want_sites, bool skip_nclass,
                        uint32_t *rec, uint32_t *rec_ent, uint32_t *site, uint32_t *site_cur, uint32_t *site_ent,
                        uint32_t n_panels, unsigned long long *total, hipStream_t stream)
{
    const unsigned blocks = (unsigned)((set.n + 31) / 32);
    if (fill)
        hipLaunchKernelGGL(index_kernel<true>, dim3(blocks), dim3(256), 0, stream, set.planes, ref_planes,
                           (uint32_t)set.n, (uint32_t)set.nchunks, (uint32_t)set.npad, want_sites ? 1 : 0,
                           skip_nclass ? 1 : 0, rec, rec_ent, site, site_cur, site_ent, n_panels, total);
    else
        hipLaunchKernelGGL(index_kernel<false>, dim3(blocks), dim3(256), 0, stream, set.planes, ref_planes,
                           (uint32_t)set.n, (uint32_t)set.nchunks, (uint32_t)set.npad, want_sites ? 1 : 0,
                           skip_nclass ? 1 : 0, rec, rec_ent, site, site_cur, site_ent, n_panels, total);
    return hipGetLastError();
}

size_t scan_tmp_words(size_t n)
{
    size_t words = 0;
    while (n > 1) {
        n = (n + kScanPerBlock - 1) / kScanPerBlock;
        words += n + 1;
    }
    return words + 2;
}

hipError_t launch_exclusive_scan(uint32_t *data, size_t n, uint32_t *tmp, hipStream_t stream)
{
    if (n == 0)
        return hipSuccess;
    const size_t nb = (n + kScanPerBlock - 1) / kScanPerBlock;
    hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nb), dim3(256), 0, stream, data, n, tmp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || nb == 1)
        return e;
    e = launch_exclusive_scan(tmp, nb, tmp + nb + 1, stream);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(256), 0, stream, data, n, tmp);
    return hipGetLastError();
}

hipError_t launch_aconst(const DeviceSet &set, const uint8_t *ref_nib, int family, bool wide,
                         const ConsensusLut *d_lut, hipStream_t stream)
{
    hipLaunchKernelGGL(aconst_kernel, dim3((unsigned)((set.n + 3) / 4)), dim3(256), 0, stream, set.rec.off,
                       set.rec.ent, ref_nib, d_lut, family, wide ? 1 : 0, family_words(family, wide),
                       (uint32_t)set.n, (uint32_t)set.npad, set.aconst);
    return hipGetLastError();
}

namespace {

template <int FAM, bool WIDE, int OUT>
hipError_t launch_cpair(const ConsensusLaunch &cl, const FWords &fw, hipStream_t stream)
{
    constexpr int W = Pack<FAM, WIDE>::W;
    const size_t smem = ((size_t)W * kPanelCols + 257 + 256 + 256 + 4) * sizeof(uint32_t);
    auto kern = consensus_pair_kernel<FAM, WIDE, OUT>;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess)
            return e;
    }
    hipLaunchKernelGGL(kern, dim3(cl.ntiles), dim3(256), smem, stream, cl.rows->rec.off, cl.rows->rec.ent,
                       cl.rows->aconst, (uint32_t)cl.rows->npad, cl.cols->site.off, cl.cols->site.ent,
                       cl.cols->aconst, (uint32_t)cl.cols->npad, cl.cols->site.n_panels, cl.cols->ref.nib, cl.d_lut,
                       fw, cl.d_tiles, cl.d_out, cl.rows->counts, cl.cols->counts, (uint32_t)cl.cols->n,
                       (uint32_t)cl.row_begin, cl.out_base, cl.square ? 1 : 0);
    return hipGetLastError();
}

template <int FAM, bool WIDE>
hipError_t launch_cpair_outputs(int measure, const ConsensusLaunch &cl, const FWords &fw, hipStream_t stream)
{
    if (cl.out_kind == DST_OUT_TALLY)
        return launch_cpair<FAM, WIDE, OUT_TALLY>(cl, fw, stream);
    if (cl.out_kind == DST_OUT_TALLY16) {
        if constexpr (!WIDE)
            return launch_cpair<FAM, WIDE, OUT_TALLY16>(cl, fw, stream);
        return hipErrorInvalidValue;
    }
    if constexpr (FAM == FAM_NHIGH)
        return launch_cpair<FAM, WIDE, OUT_INT>(cl, fw, stream);
    else if constexpr (FAM == FAM_RAW)
        return measure == DST_RAW ? launch_cpair<FAM, WIDE, DST_RAW>(cl, fw, stream)
                                  : launch_cpair<FAM, WIDE, DST_JC69>(cl, fw, stream);
    else if constexpr (FAM == FAM_K80)
        return launch_cpair<FAM, WIDE, DST_K80>(cl, fw, stream);
    else
        return launch_cpair<FAM, WIDE, DST_TN93>(cl, fw, stream);
}

}  // namespace

hipError_t launch_consensus_pairs(int measure, const ConsensusLaunch &cl, const uint32_t f_words[kMaxWords],
                                  hipStream_t stream)
{
    FWords fw;
    for (int w = 0; w < kMaxWords; ++w)
        fw.w[w] = f_words[w];
    const int fam = family_of(measure);
#define DST_FAM(F)                                                                     \
    case F:                                                                            \
        return cl.wide ? launch_cpair_outputs<F, true>(measure, cl, fw, stream)        \
                       : launch_cpair_outputs<F, false>(measure, cl, fw, stream);
    switch (fam) {
        DST_FAM(FAM_NHIGH)
        DST_FAM(FAM_RAW)
        DST_FAM(FAM_K80)
        DST_FAM(FAM_TN93)
    default: break;
    }
#undef DST_FAM
    return hipErrorInvalidValue;
}

hipError_t launch_site_hist(const DeviceSet &set, uint32_t *hist, hipStream_t stream)
{
    const size_t blocks = set.nchunks * ((set.n + kHistRecords - 1) / kHistRecords);
    if (blocks > 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(site_hist_kernel, dim3((unsigned)blocks), dim3(128), 0, stream, reinterpret_cast<const uint32_t *>(set.planes),
                       (uint32_t)set.n, (uint32_t)set.len, (uint32_t)set.nchunks, (uint32_t)set.npad, hist);
    return hipGetLastError();
}

}  // namespace dst
