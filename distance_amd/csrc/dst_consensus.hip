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
//   ref_sample_kernel   per-site plurality code over a sample of records -> the reference c (from the planes, or
//                       from the byte matrix BEFORE the pack, which then counts the list lengths on its way)
//   index_kernel        per record: ascending list of (site, nibble) where it differs from c (from the planes)
//   slot_fill_kernel    the same lists from the 16-byte slots the pack left per (record, chunk); both mark where a list
//                       crosses every 1,024th site (range_start), and chunks with many differences (runs of N) are
//                       written by a whole wave (emit_chunk_by_wave)
//   sum2 / report       list totals; what the host reads after an upload, written into page-locked memory
//   scan kernels        exclusive scan of the list lengths -> CSR offsets
//   site_bucket_kernel  the same entries by (panel of 2,048 column records, site): 32-byte lookup-table entries that
//                       hold the bucket itself, assembled per (panel, 1,024 sites) in LDS from the lists' pieces
//   aconst_kernel       A_k(record), packed like the accumulators
//   consensus_pair_kernel  one block = a few rows x one column panel: the rows' lists are joined with the
//                       site buckets of the panel, h_k goes into LDS accumulators (ds_add_u32), then one
//                       coalesced pass adds the per-record constants, finalises (f64, reference operation
//                       order) and stores in canonical order (nontemporal: written once, never read)
//   site_hist_kernel    exact per-site base counts for consensus() itself (dst_consensus)
//
// Integer adds only (order-independent, exact); two 16-bit tallies share one 32-bit accumulator when the
// alignment is shorter than 65,536 sites (the sums are exact modulo 2^32 and every final tally fits).
#include "dst_device.hpp"

#ifndef DST_OVF_DEPTH
#define DST_OVF_DEPTH 2   // measured 1 / 2 / 4 / 8 / 16: 21.1 / 17.7 / 17.3 / 24.2 / 25.6 ms on the N-heavy case of tools/nrun_bench.py
#endif

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
// One block = one 128-site chunk: 4 groups of 128 threads, thread = (site, every 4th sample); the groups'
// class counts meet in LDS.  Every thread of a group reads the same 16 bytes of a sampled record (broadcast).
// Classes: known A, G, C, T and the N class (N, -, ?); ties go to the first in that order.  Any choice
// gives exact results — the reference only decides how much work the pair kernel has.
// record of sample k: floor(k n / samples) without a 64-bit division (samples is kRefSamples = 512, or n when n is smaller)
__device__ __forceinline__ uint32_t sample_record(uint32_t k, uint32_t n, uint32_t samples)
{
    static_assert(kRefSamples == 512, "shifts below");
    return samples == n ? k : k * (n >> 9) + ((k * (n & 511u)) >> 9);
}

// ascending list of the hot sites (one block; nchunks is at most a few ten thousand)
__global__ __launch_bounds__(1024) void hot_list_kernel(const uint4 *__restrict__ hot_planes, uint32_t nchunks,
                                                        uint32_t *__restrict__ hot_sites,
                                                        const uint32_t *__restrict__ partials, uint32_t samples,
                                                        unsigned long long *__restrict__ stats)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t base;
    __shared__ unsigned long long sums[16][8];
    {   // the sample's statistics: the chunks' shares (ref_sample_kernel) summed
        unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t c = threadIdx.x; c < 2 * nchunks; c += 1024)   // two shares per chunk
#pragma unroll
            for (int k = 0; k < 8; ++k)
                acc[k] += partials[(size_t)c * 8 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
                acc[k] += __shfl_xor(acc[k], o);
            if ((threadIdx.x & 63u) == 0)
                sums[threadIdx.x >> 6][k] = acc[k];
        }
    }
    if (threadIdx.x == 0)
        base = 0;
    __syncthreads();
    if (threadIdx.x < 8) {
        unsigned long long t = 0;
        for (int wv = 0; wv < 16; ++wv)
            t += sums[wv][threadIdx.x];
        stats[threadIdx.x] = threadIdx.x == 3 ? samples : t;
    }
    for (uint32_t c0 = 0; c0 < nchunks; c0 += 1024) {
        const uint32_t c = c0 + threadIdx.x;
        const uint4 m = c < nchunks ? hot_planes[c] : make_uint4(0, 0, 0, 0);
        const uint32_t pc = popc4(m);
        uint32_t incl = pc, up;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            up = __shfl_up(incl, o);
            if ((threadIdx.x & 63u) >= (uint32_t)o) incl += up;
        }
        if ((threadIdx.x & 63u) == 63u)
            wave_tot[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t off = base, total = 0;
        for (uint32_t wv = 0; wv < 16; ++wv) {
            if (wv < (threadIdx.x >> 6)) off += wave_tot[wv];
            total += wave_tot[wv];
        }
        uint32_t at = off + incl - pc;
        const uint32_t w4[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t bits = w4[w];
            while (bits) {
                const uint32_t bit = (uint32_t)__builtin_ctz(bits);
                bits &= bits - 1;
                hot_sites[at++] = c * kChunkSites + 32u * w + bit;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0)
            base += total;
        __syncthreads();
    }
}

// BYTES: the sample is read from the row-major code matrix itself (high nibble = the A,G,C,T plane bits), so the
// reference exists BEFORE the pack and the pack can count every record's differences on its way (dst_kernels.hip).
template <bool BYTES>
__global__ __launch_bounds__(BYTES ? 1024 : 512) void ref_sample_kernel(const uint32_t *__restrict__ planes32,
                                                         const uint8_t *__restrict__ codes, size_t row_stride, uint32_t n,
                                                         uint32_t len, uint32_t nchunks, uint32_t npad,
                                                         uint32_t samples, uint4 *__restrict__ ref_planes,
                                                         uint4 *__restrict__ hot_planes,
                                                         uint32_t *__restrict__ partials,
                                                         uint32_t *__restrict__ zero, uint32_t zero_words,
                                                         unsigned long long *__restrict__ first_bad)
{
    // what the pack behind this kernel adds to / takes the minimum of, cleared here instead of by two fills of their own
    if constexpr (BYTES) {
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < zero_words; i += gridDim.x * blockDim.x)
            zero[i] = 0;
        if (first_bad && blockIdx.x == 0 && threadIdx.x == 0)
            *first_bad = ~0ull;
    }
    // GROUPS groups of 128 threads, thread = (site, every GROUPS-th sample), UNR sampled records per round: their
    // loads are issued together (one record at a time this kernel was a chain of 128 memory latencies on one block
    // per CU: 0.12 ms for 8 MB)
    constexpr uint32_t GROUPS = BYTES ? 8 : 4, UNR = BYTES ? 16 : 8;
    __shared__ uint32_t part[GROUPS][5][128];
    const uint32_t c = blockIdx.x, b = threadIdx.x & 127u, grp = threadIdx.x >> 7;
    const uint32_t w = b >> 5, bit = b & 31;
    const size_t ps = (size_t)nchunks * npad * 4;  // plane stride in 32-bit words
    uint32_t cnt[5] = {0, 0, 0, 0, 0};
    for (uint32_t k0 = grp; k0 < samples; k0 += GROUPS * UNR) {
        uint32_t nibs[UNR];
        if constexpr (BYTES) {
            uint32_t by[UNR];
            const uint32_t site = c * kChunkSites + b;
#pragma unroll
            for (uint32_t u = 0; u < UNR; ++u) {
                const uint32_t k = k0 + GROUPS * u;
                const uint32_t r = sample_record(min(k, samples - 1), n, samples);
                by[u] = codes[(size_t)r * row_stride + min(site, len - 1u)];   // (unconditional: all in flight together)
            }
            if (site >= len) {
#pragma unroll
                for (uint32_t u = 0; u < UNR; ++u)
                    by[u] = 0xF0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < UNR; ++u)
                nibs[u] = by[u] >> 4;
        } else {
            uint32_t pw[UNR][4];
#pragma unroll
            for (uint32_t u = 0; u < UNR; ++u) {
                const uint32_t k = k0 + GROUPS * u;
                const uint32_t r = sample_record(min(k, samples - 1), n, samples);
                const size_t at = ((size_t)c * npad + r) * 4 + w;
                pw[u][0] = planes32[PL_A * ps + at];
                pw[u][1] = planes32[PL_G * ps + at];
                pw[u][2] = planes32[PL_C * ps + at];
                pw[u][3] = planes32[PL_T * ps + at];
            }
#pragma unroll
            for (uint32_t u = 0; u < UNR; ++u)
                nibs[u] = ((pw[u][0] >> bit) & 1u) << 3 | ((pw[u][1] >> bit) & 1u) << 2 |
                          ((pw[u][2] >> bit) & 1u) << 1 | ((pw[u][3] >> bit) & 1u);
        }
#pragma unroll
        for (uint32_t u = 0; u < UNR; ++u) {
            if (k0 + GROUPS * u >= samples)
                break;
            const uint32_t nib = nibs[u];
            cnt[0] += nib == 8;
            cnt[1] += nib == 4;
            cnt[2] += nib == 2;
            cnt[3] += nib == 1;
            cnt[4] += nib == 15;
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
        part[grp][k][b] = cnt[k];
    __syncthreads();
    if (grp == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        cnt[k] = 0;
#pragma unroll
        for (uint32_t g = 0; g < GROUPS; ++g)
            cnt[k] += part[g][k][b];
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
    const uint32_t wave = b >> 6;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned long long m = __ballot((nib >> (3 - p)) & 1u);
        if ((b & 63) == 0)
            reinterpret_cast<unsigned long long *>(&ref_planes[(size_t)p * nchunks + c])[wave] = m;
    }
    // "hot" sites: more than kHotPermille of the sampled records deviate from the plurality (clade-defining
    // mutations, indel-rich columns).  Their events grow with p^2; the hybrid path gives these columns to the
    // dense kernels instead (dst_api.cpp) and keeps them out of the lists.
    const uint32_t devs = real ? samples - best : 0u;
    const bool hot = devs * 1000u > samples * kHotPermille;
    const unsigned long long hot_mask = __ballot(hot);
    if ((b & 63) == 0)
        reinterpret_cast<unsigned long long *>(&hot_planes[c])[wave] = hot_mask;
    // statistics for the path choice: known reference sites, sum and sum of squares of the sampled
    // records that deviate from the plurality class — over all sites and over the cold ones alone
    const unsigned long long known = __ballot(real && cls < 4);
    const unsigned long long known_hot = __ballot(real && cls < 4 && hot);
    uint32_t dev = devs, dev2 = devs * devs, cdev = hot ? 0u : devs, cdev2 = hot ? 0u : devs * devs;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dev += __shfl_xor(dev, o);
        dev2 += __shfl_xor(dev2, o);
        cdev += __shfl_xor(cdev, o);
        cdev2 += __shfl_xor(cdev2, o);
    }
    // this wave's share of the statistics (two per chunk, summed by hot_list_kernel: no same-address atomics)
    if ((b & 63u) == 0) {
        const uint32_t mine[8] = {(uint32_t)__builtin_popcountll(known), dev, dev2, 0u, (uint32_t)__builtin_popcountll(hot_mask),
                                  (uint32_t)__builtin_popcountll(known_hot), cdev, cdev2};
#pragma unroll
        for (int k = 0; k < 8; ++k)
            partials[((size_t)c * 2 + wave) * 8 + k] = mine[k];
    }
    }   // grp == 0
    // (measured and not kept: the statistics' sum and the hot-site list as the tail of whichever block finishes last —
    // the agent-scope fences that needs, an L2 write-back per block on this multi-die part, cost 60 us against the
    // 13 us of hot_list_kernel as a launch of its own)
}

// The hot columns of a set as a packed set of their own (all 8 planes): one thread = one (record, chunk of 128
// hot sites); lanes run along records like pack_kernel's, the site list is wave-uniform.
__global__ __launch_bounds__(256) void compact_kernel(const uint32_t *__restrict__ planes32, uint32_t n, uint32_t nchunks,
                                                      uint32_t npad, const uint32_t *__restrict__ hot_sites, uint32_t n_hot,
                                                      uint32_t hot_chunks, uint32_t hot_npad, uint4 *__restrict__ out)
{
    const uint32_t r = blockIdx.y * blockDim.x + threadIdx.x;
    const uint32_t hc = blockIdx.x;
    if (r >= hot_npad)
        return;
    uint32_t o[PL_COUNT][4];
#pragma unroll
    for (int p = 0; p < PL_COUNT; ++p)
#pragma unroll
        for (int w = 0; w < 4; ++w)
            o[p][w] = (p <= PL_T) ? 0xFFFFFFFFu : 0u;  // all N
    if (r < n) {
        const size_t ps = (size_t)nchunks * npad * 4;
        for (uint32_t k = 0; k < kChunkSites; ++k) {
            const uint32_t idx = hc * kChunkSites + k;
            if (idx >= n_hot)
                break;
            const uint32_t s = hot_sites[idx];
            const size_t at = ((size_t)(s >> 7) * npad + r) * 4 + ((s >> 5) & 3u);
            const uint32_t bit = s & 31u, ow = k >> 5, ob = k & 31u;
#pragma unroll
            for (int p = 0; p <= PL_T; ++p) {   // the base planes; the source set may be lean (no other planes)
                const uint32_t v = (planes32[p * ps + at] >> bit) & 1u;
                o[p][ow] = (o[p][ow] & ~(1u << ob)) | (v << ob);
            }
        }
        // K, X1, X0, CL of the gathered columns from their base bits (what pack_kernel's split4 computes)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t A = o[PL_A][w], G = o[PL_G][w], C = o[PL_C][w], T = o[PL_T][w];
            const uint32_t pur = A | G, pyr = C | T;
            const uint32_t K = (A ^ G ^ C ^ T) & ~((A & G) | (C & T));
            const uint32_t X1 = pyr & ~pur;
            o[PL_K][w] = K;
            o[PL_X1][w] = X1;
            o[PL_CL][w] = X1 | (pur & ~pyr);
            o[PL_X0][w] = K & (G | T);
        }
    }
#pragma unroll
    for (int p = 0; p < PL_COUNT; ++p)
        out[((size_t)p * hot_chunks + hc) * hot_npad + r] = make_uint4(o[p][0], o[p][1], o[p][2], o[p][3]);
}

// The entries of ONE (record rs, chunk cs) written by the whole wave, two sites per lane, from `out0` on in ascending
// site order.  For chunks with many differences (a run of N, a gap): left to its own lane such a chunk is a serial loop
// over up to 128 sites, and the longest run of one record then sets the whole kernel's time.  Wave-uniform arguments.
__device__ __forceinline__ void emit_chunk_by_wave(const uint4 *__restrict__ planes, const uint4 *__restrict__ ref_planes,
                                                   const uint4 *__restrict__ hot_planes, bool skip_nclass, size_t ps,
                                                   uint32_t nchunks, uint32_t npad, uint32_t rs, uint32_t cs, uint32_t out0,
                                                   uint32_t lane, uint32_t *__restrict__ rec_ent, uint32_t ent_cap = 0xFFFFFFFFu)
{
    const uint32_t w = lane >> 4, b0 = (lane & 15u) * 2u;
    const size_t wa = ((size_t)cs * npad + rs) * 4u + w;
    const uint32_t *const pw = reinterpret_cast<const uint32_t *>(planes);
    const uint32_t *const rw = reinterpret_cast<const uint32_t *>(ref_planes);
    const uint32_t a = pw[PL_A * ps * 4u + wa], g = pw[PL_G * ps * 4u + wa], cc = pw[PL_C * ps * 4u + wa], t = pw[PL_T * ps * 4u + wa];
    const uint32_t ra = rw[(size_t)cs * 4u + w], rg = rw[((size_t)nchunks + cs) * 4u + w],
                   rc = rw[(2 * (size_t)nchunks + cs) * 4u + w], rt = rw[(3 * (size_t)nchunks + cs) * 4u + w];
    uint32_t dm = (a ^ ra) | (g ^ rg) | (cc ^ rc) | (t ^ rt);
    if (hot_planes)
        dm &= ~reinterpret_cast<const uint32_t *>(hot_planes)[(size_t)cs * 4u + w];
    if (skip_nclass)
        dm &= ~(a & g & cc & t);
    // entries before this lane's sites: the words below + the bits below in its own word
    const uint32_t pcw = (uint32_t)__builtin_popcount(dm);
    const uint32_t p0 = __shfl(pcw, 0), p1 = __shfl(pcw, 16), p2 = __shfl(pcw, 32);
    uint32_t pos = out0 + (w > 0 ? p0 : 0u) + (w > 1 ? p1 : 0u) + (w > 2 ? p2 : 0u) +
                   (uint32_t)__builtin_popcount(dm & ((1u << b0) - 1u));
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k) {
        const uint32_t bit = b0 + k;
        if (dm >> bit & 1u) {
            const uint32_t nib = ((a >> bit) & 1u) << 3 | ((g >> bit) & 1u) << 2 | ((cc >> bit) & 1u) << 1 | ((t >> bit) & 1u);
            const uint32_t rnib = ((ra >> bit) & 1u) << 3 | ((rg >> bit) & 1u) << 2 | ((rc >> bit) & 1u) << 1 | ((rt >> bit) & 1u);
            if (pos < ent_cap)
                rec_ent[pos] = (cs * kChunkSites + 32u * w + bit) | (uint32_t)ref_class(rnib) << kSiteBits | nib << kEntryShift;
            ++pos;
        }
    }
}

// =============================================================================================
// difference lists
// =============================================================================================
// One wave = 8 records x 8 chunks per step (lane = chunk-lane * 8 + record-lane), so the eight lanes of
// a chunk read one 128-byte line of each plane, and a record's entries come out in ascending site order:
// the exclusive prefix over the chunk-lanes of a record is three shuffles.
// FILL == false: rec[r] = list length.   FILL == true: rec = scanned offsets, entries written in ascending site order.
// (A column set's lists and buckets are filled together by site_fill_kernel below instead.)
template <bool FILL>
__global__ __launch_bounds__(256) void index_kernel(const uint4 *__restrict__ planes,
                                                    const uint4 *__restrict__ ref_planes,
                                                    const uint4 *__restrict__ hot_planes, uint32_t n,
                                                    uint32_t nchunks, uint32_t npad, int skip_nclass,
                                                    uint32_t *__restrict__ rec, uint32_t *__restrict__ rec_ent,
                                                    unsigned long long *__restrict__ total, uint32_t *__restrict__ range_start)
{
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t rl = lane & 7u, cl = lane >> 3;
    const uint32_t r = wave * 8u + rl;
    const bool live = r < n;
    const size_t ps = (size_t)nchunks * npad;
    uint32_t run = 0;
    const uint32_t base0 = (FILL && live) ? rec[r] : 0u;
    // the planes of the NEXT step are loaded before this step's entries are written out (a wave walks its records'
    // chunks one step after the other: without this every step waits a full memory latency)
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    uint4 nA = zero4, nG = zero4, nC = zero4, nT = zero4;
    auto load_step = [&](uint32_t c0) {
        const uint32_t c = c0 + cl;
        if (live && c < nchunks) {
            const size_t at = (size_t)c * npad + r;
            nA = planes[PL_A * ps + at];
            nG = planes[PL_G * ps + at];
            nC = planes[PL_C * ps + at];
            nT = planes[PL_T * ps + at];
        }
    };
    load_step(0);
    for (uint32_t c0 = 0; c0 < nchunks; c0 += 8) {
        const uint32_t c = c0 + cl;
        uint4 A = nA, G = nG, C = nC, T = nT, d = zero4, rA = zero4, rG = zero4, rC = zero4, rT = zero4;
        if (c0 + 8 < nchunks)
            load_step(c0 + 8);
        if (live && c < nchunks) {
            rA = ref_planes[c];
            rG = ref_planes[nchunks + c];
            rC = ref_planes[2 * (size_t)nchunks + c];
            rT = ref_planes[3 * (size_t)nchunks + c];
            d.x = (A.x ^ rA.x) | (G.x ^ rG.x) | (C.x ^ rC.x) | (T.x ^ rT.x);
            d.y = (A.y ^ rA.y) | (G.y ^ rG.y) | (C.y ^ rC.y) | (T.y ^ rT.y);
            d.z = (A.z ^ rA.z) | (G.z ^ rG.z) | (C.z ^ rC.z) | (T.z ^ rT.z);
            d.w = (A.w ^ rA.w) | (G.w ^ rG.w) | (C.w ^ rC.w) | (T.w ^ rT.w);
            if (hot_planes) {  // hybrid path: the hot columns belong to the dense kernels
                const uint4 h = hot_planes[c];
                d.x &= ~h.x;
                d.y &= ~h.y;
                d.z &= ~h.z;
                d.w &= ~h.w;
            }
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
        const bool by_wave = FILL && pc > kSlotEntries;   // many differences in one chunk: the whole wave writes them
        if constexpr (FILL) {
            const uint32_t at0 = base0 + run + (incl - pc);
            static_assert(kBucketSites == 8 * kChunkSites, "a step of this kernel is one range of site_bucket_kernel");
            if (range_start && live && cl == 0)
                range_start[(size_t)(c0 >> 3) * npad + r] = at0;
            for (unsigned long long todo = __ballot(by_wave); todo;) {
                const uint32_t src = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1;
                emit_chunk_by_wave(planes, ref_planes, hot_planes, skip_nclass != 0, ps, nchunks, npad, wave * 8u + (src & 7u),
                                   c0 + (src >> 3), __shfl(at0, src), lane, rec_ent);
            }
        }
        if (pc && !by_wave) {
            uint32_t at = base0 + run + (incl - pc);
            const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
            const uint32_t aw[4] = {A.x, A.y, A.z, A.w}, gw[4] = {G.x, G.y, G.z, G.w};
            const uint32_t cw[4] = {C.x, C.y, C.z, C.w}, tw[4] = {T.x, T.y, T.z, T.w};
            const uint32_t raw[4] = {rA.x, rA.y, rA.z, rA.w}, rgw[4] = {rG.x, rG.y, rG.z, rG.w};
            const uint32_t rcw[4] = {rC.x, rC.y, rC.z, rC.w}, rtw[4] = {rT.x, rT.y, rT.z, rT.w};
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
                        const uint32_t rnib = ((raw[w] >> bit) & 1u) << 3 | ((rgw[w] >> bit) & 1u) << 2 |
                                              ((rcw[w] >> bit) & 1u) << 1 | ((rtw[w] >> bit) & 1u);
                        rec_ent[at++] = s | (uint32_t)ref_class(rnib) << kSiteBits | nib << kEntryShift;
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

// The fill pass from the pack's slots: same output as index_kernel<true> (ascending site order), but a (record, chunk)
// is one 16-byte slot that already holds its entries — a quarter of the four planes' bytes.  A chunk with more than
// kSlotEntries differences goes back to the planes.
// A wave = (64 / CLN) records x CLN chunks per step.  8 x 8 reads 128 contiguous bytes per chunk (what a big set
// wants); a small set has too few waves that way, each a chain of nchunks / 8 steps of ~400 instructions (10,000 x
// 30,000: 1,250 waves on 1,024 SIMDs, 78 us), so it takes 2 records x 32 chunks: four times the waves, a quarter of
// the steps.  WITHOUT_HOT (the hybrid path's lists) filters entries; without it an entry's place is a constant offset.
template <uint32_t CLN, bool WITHOUT_HOT>
__global__ __launch_bounds__(256) void slot_fill_kernel(const uint4 *__restrict__ slots, const uint4 *__restrict__ planes,
                                                        const uint4 *__restrict__ ref_planes,
                                                        const uint4 *__restrict__ hot_planes, uint32_t n,
                                                        uint32_t nchunks, uint32_t npad,
                                                        const uint32_t *__restrict__ rec_off, uint32_t *__restrict__ rec_ent,
                                                        uint32_t *__restrict__ range_start, uint32_t rec_first, uint32_t ent_cap,
                                                        const uint32_t *__restrict__ run_index, const uint32_t *__restrict__ run_state)
{
    // records [rec_first, n) (rec_off[0] is record rec_first's offset); entries at or beyond ent_cap are not written:
    // one rank's share of a set, whose lists go into an exchange block of fixed size (dst_upload_shared)
    constexpr uint32_t RLN = 64u / CLN;
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t rl = lane % RLN, cl = lane / RLN;
    const uint32_t r = rec_first + wave * RLN + rl;
    const bool live = r < n;
    const size_t ps = (size_t)nchunks * npad;
    uint32_t run = 0;
    const uint32_t base0 = live ? rec_off[r - rec_first] : 0u;
    // a run record (RunIndex: kRunMin and more chunks of N) leaves its run chunks out of its list
    const bool run_record = live && run_state && run_state[1] != 0 && run_index[r] != 0xFFFFFFFFu;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    // two steps ahead
    uint4 nslot = (live && cl < nchunks) ? slots[(size_t)cl * npad + r] : zero4;
    uint4 nslot2 = (live && cl + CLN < nchunks) ? slots[(size_t)(cl + CLN) * npad + r] : zero4;
    for (uint32_t c0 = 0; c0 < nchunks; c0 += CLN) {
        const uint32_t c = c0 + cl;
        const uint4 slot = nslot;
        nslot = nslot2;
        if (c0 + 2 * CLN < nchunks)
            nslot2 = (live && c + 2 * CLN < nchunks) ? slots[(size_t)(c + 2 * CLN) * npad + r] : zero4;
        const uint32_t sw[4] = {slot.x, slot.y, slot.z, slot.w};
        const bool run_chunk = (slot.x >> 8) & 1u;                 // 128 sites of N: nothing inline in the slot
        const uint32_t cnt_all = run_chunk && run_record ? 0u : slot.x & 0xFFu;
        const bool big = cnt_all > kSlotEntries || (run_chunk && cnt_all != 0);
        // entries this lane emits: from the slot, or (big) from the planes
        uint32_t pc = 0;
        if (big) {
            const size_t at = (size_t)c * npad + r;
            const uint4 A = planes[PL_A * ps + at], G = planes[PL_G * ps + at], C = planes[PL_C * ps + at], T = planes[PL_T * ps + at];
            const uint4 rA = ref_planes[c], rG = ref_planes[nchunks + c], rC = ref_planes[2 * (size_t)nchunks + c],
                        rT = ref_planes[3 * (size_t)nchunks + c];
            uint4 d;
            d.x = (A.x ^ rA.x) | (G.x ^ rG.x) | (C.x ^ rC.x) | (T.x ^ rT.x);
            d.y = (A.y ^ rA.y) | (G.y ^ rG.y) | (C.y ^ rC.y) | (T.y ^ rT.y);
            d.z = (A.z ^ rA.z) | (G.z ^ rG.z) | (C.z ^ rC.z) | (T.z ^ rT.z);
            d.w = (A.w ^ rA.w) | (G.w ^ rG.w) | (C.w ^ rC.w) | (T.w ^ rT.w);
            if (WITHOUT_HOT) {
                const uint4 h = hot_planes[c];
                d.x &= ~h.x;
                d.y &= ~h.y;
                d.z &= ~h.z;
                d.w &= ~h.w;
            }
            pc = popc4(d);
        } else if (WITHOUT_HOT) {
#pragma unroll
            for (uint32_t k = 1; k <= kSlotEntries; ++k) {
                const uint32_t e = (sw[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu;
                pc += (k <= cnt_all && !(e >> 14 & 1u)) ? 1u : 0u;
            }
        } else {
            pc = cnt_all;
        }
        // inclusive scan over the chunk lanes of a record (lanes RLN apart)
        uint32_t incl = pc;
#pragma unroll
        for (uint32_t o = RLN; o < 64u; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        const uint32_t tot = __shfl(incl, 64u - RLN + rl);
        const uint32_t at0 = base0 + run + (incl - pc);
        // where the record's list crosses a multiple of kBucketSites (site_bucket_kernel's pieces)
        if (range_start && live && c < nchunks && c % (kBucketSites / kChunkSites) == 0)
            range_start[(size_t)(c / (kBucketSites / kChunkSites)) * npad + r] = at0;
        // a chunk that did not fit its slot is emitted by the whole wave (50 of 72 us at 10,000 x 30,000 were the serial
        // loops of the few lanes holding a run of N)
        for (unsigned long long todo = __ballot(big && pc != 0); todo;) {
            const uint32_t src = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            emit_chunk_by_wave(planes, ref_planes, WITHOUT_HOT ? hot_planes : nullptr, false, ps, nchunks, npad,
                               rec_first + wave * RLN + src % RLN, c0 + src / RLN, __shfl(at0, src), lane, rec_ent, ent_cap);
        }
        if (pc && !big) {
            uint32_t at = at0;
            if (WITHOUT_HOT) {
#pragma unroll
                for (uint32_t k = 1; k <= kSlotEntries; ++k) {
                    const uint32_t e = (sw[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu;
                    if (k <= cnt_all && !(e >> 14 & 1u)) {
                        if (at < ent_cap)
                            rec_ent[at] = (c * kChunkSites + (e & 127u)) | ((e >> 7) & 7u) << kSiteBits | ((e >> 10) & 15u) << kEntryShift;
                        ++at;
                    }
                }
            } else {
                // entry k of the slot goes to at + k - 1: constant offsets from one address
                uint32_t *const ep = rec_ent + at;
                const uint32_t site0 = c * kChunkSites;
#pragma unroll
                for (uint32_t k = 1; k <= kSlotEntries; ++k) {
                    const uint32_t e = (sw[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu;
                    if (k <= cnt_all && at + k - 1 < ent_cap) {
                        ep[k - 1] = (site0 + (e & 127u)) | ((e >> 7) & 7u) << kSiteBits | ((e >> 10) & 15u) << kEntryShift;
                    }
                }
            }
        }
        run += tot;
    }
}

// =============================================================================================
// site buckets of a column set, from its difference lists
// =============================================================================================
// One block = one (panel of 2,048 records, range of kBucketSites sites): it owns the buckets of that piece, so
// their sizes are LDS counters and their 32-byte lookup-table entries are assembled in LDS and leave as one
// contiguous 32 KB piece — no global atomics, no scattered 2-byte writes into the table.  A record's list is in
// ascending site order and the fill pass notes where it crosses every multiple of kBucketSites (range_start), so a
// thread knows the piece of its record's list that lies in the block's sites (r02 first streamed ALL entries of the
// panel through every block and kept 1 in 30: 16 dependent rounds of loads, 42 us a block) and walks up to kBucketWalk
// entries of it; what a long piece (a run of N) has beyond those is shared out over a wave.
//   table entry = 16 halfwords: [0] entries in the bucket (<= 2,048), [1..15] entries as record-in-panel | nibble << 11;
//   a bucket of more than kInlineEvents entries keeps kInlineOverflowing of them inline, its last word is the place
//   of the others in site_ent (record | nibble << 28).  Such buckets take a second pass over the lists, after the
//   block has counted them and reserved room for all its overflow entries with ONE global atomic.
constexpr uint32_t kBucketThreads = 1024;

constexpr uint32_t kBucketWalk = 8;
__global__ __launch_bounds__(kBucketThreads) void site_bucket_kernel(const uint32_t *__restrict__ rec_off,
                                                                     const uint32_t *__restrict__ rec_ent,
                                                                     const uint32_t *__restrict__ range_start, uint32_t n,
                                                                     uint32_t npad, uint32_t n_sites, uint4 *__restrict__ site_inl,
                                                                     uint32_t *__restrict__ site_ent,
                                                                     uint32_t *__restrict__ ovf_total)
{
    constexpr uint32_t NT = kBucketThreads, RPT = kPanelCols / NT;   // records per thread
    static_assert(NT == kBucketSites, "one bucket per thread in the offset scan");
    static_assert(RPT * NT == kPanelCols, "every record of the panel has its thread");
    __shared__ uint32_t cnt[kBucketSites];
    __shared__ uint32_t ooff[kBucketSites];
    __shared__ __attribute__((aligned(16))) uint16_t tabl[kBucketSites][16];
    __shared__ uint32_t wave_tot[NT / 64];
    __shared__ uint32_t blk_base;
    __shared__ uint32_t long_n, long_from[kPanelCols], long_to[kPanelCols];
    __shared__ uint16_t long_rec[kPanelCols];
    const uint32_t panel = blockIdx.y, tid = threadIdx.x, lane = tid & 63u;
    const uint32_t s0 = blockIdx.x * kBucketSites, ns = min(kBucketSites, n_sites - s0);
    const uint32_t r0 = panel * kPanelCols, nrec = min(kPanelCols, n - r0);
    cnt[tid] = 0;
    if (tid == 0)
        long_n = 0;
    for (uint32_t k = tid; k < kBucketSites * 2; k += NT)
        reinterpret_cast<uint4 *>(&tabl[0][0])[k] = make_uint4(0, 0, 0, 0);
    // the piece of each of this thread's records' lists that lies in the block's sites: the fill pass noted where every
    // record's list crosses a multiple of kBucketSites (range_start) — no search, and the piece's length is known
    uint32_t from[RPT], to[RPT];
#pragma unroll
    for (uint32_t k = 0; k < RPT; ++k) {
        const uint32_t rr = tid + k * NT;
        from[k] = to[k] = 0;
        if (rr < nrec) {
            from[k] = range_start[(size_t)blockIdx.x * npad + r0 + rr];
            to[k] = blockIdx.x + 1 < gridDim.x ? range_start[(size_t)(blockIdx.x + 1) * npad + r0 + rr] : rec_off[r0 + rr + 1];
        }
    }
    __syncthreads();
    // every list entry of the panel whose site is in this block's range: visit(site - s0, record in panel, nibble).
    // first == true also notes the pieces that go on beyond a thread's kBucketWalk entries (the second pass reuses the notes)
    auto for_each_entry = [&](bool first, auto &&visit) {
#pragma unroll
        for (uint32_t k = 0; k < RPT; ++k) {
            const uint32_t rr = tid + k * NT;
            uint32_t e[kBucketWalk];
#pragma unroll
            for (uint32_t j = 0; j < kBucketWalk; ++j)   // (unconditional loads from a clamped index: all in flight together)
                e[j] = rec_ent[min(from[k] + j, to[k] ? to[k] - 1u : 0u)];
#pragma unroll
            for (uint32_t j = 0; j < kBucketWalk; ++j)
                if (from[k] + j < to[k])
                    visit((e[j] & kSiteMask) - s0, rr, e[j] >> kEntryShift);
            if (first && from[k] + kBucketWalk < to[k]) {
                const uint32_t q = atomicAdd(&long_n, 1u);
                long_rec[q] = (uint16_t)rr;
                long_from[q] = from[k] + kBucketWalk;
                long_to[q] = to[k];
            }
        }
        __syncthreads();
        const uint32_t nq = long_n;
        for (uint32_t q = tid >> 6; q < nq; q += NT / 64) {   // a wave per piece: the pieces' loads overlap
            const uint32_t rr = long_rec[q], hi = long_to[q];
            for (uint32_t j = long_from[q] + lane; j < hi; j += 64u) {
                const uint32_t ev = rec_ent[j];
                visit((ev & kSiteMask) - s0, rr, ev >> kEntryShift);
            }
        }
    };
    // pass 1: sizes, and the entries of every bucket as if it fitted
    for_each_entry(true, [&](uint32_t sl, uint32_t col, uint32_t nib) {
        const uint32_t pos = atomicAdd(&cnt[sl], 1u);
        if (pos < kInlineEvents)
            tabl[sl][1 + pos] = (uint16_t)(col | nib << 11);
    });
    __syncthreads();
    // room for the overflow entries of this block's buckets: exclusive scan over the buckets + one global atomic
    const uint32_t c = cnt[tid];
    const uint32_t over = c > kInlineEvents ? c - kInlineOverflowing : 0u;
    uint32_t incl = over, up;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        up = __shfl_up(incl, o);
        if (lane >= (uint32_t)o) incl += up;
    }
    if (lane == 63u)
        wave_tot[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t wv = 0; wv < NT / 64; ++wv) {
        if (wv < (tid >> 6)) before += wave_tot[wv];
        total += wave_tot[wv];
    }
    if (tid == 0 && total)
        blk_base = atomicAdd(ovf_total, total);
    tabl[tid][0] = (uint16_t)c;   // at most kPanelCols: fits the halfword
    cnt[tid] = 0;
    __syncthreads();
    if (total) {   // (uniform over the block)
        ooff[tid] = blk_base + before + incl - over;
        if (over)
            *reinterpret_cast<uint32_t *>(&tabl[tid][14]) = ooff[tid];
        __syncthreads();
        // pass 2: the buckets that do not fit, again: kInlineOverflowing entries inline, the others to their place
        for_each_entry(false, [&](uint32_t sl, uint32_t col, uint32_t nib) {
            if (tabl[sl][0] <= kInlineEvents)
                return;
            const uint32_t pos = atomicAdd(&cnt[sl], 1u);
            if (pos < kInlineOverflowing)
                tabl[sl][1 + pos] = (uint16_t)(col | nib << 11);
            else
                site_ent[ooff[sl] + pos - kInlineOverflowing] = (r0 + col) | nib << kEntryShift;
        });
        __syncthreads();
    }
    for (uint32_t k = tid; k < ns * 2; k += NT)
        site_inl[((size_t)panel * n_sites + s0) * 2 + k] = reinterpret_cast<const uint4 *>(&tabl[0][0])[k];
}

// =============================================================================================
// exclusive scan (list lengths -> offsets)
// =============================================================================================
constexpr uint32_t kScanPerBlock = 2048;  // 256 threads x 8

// src0 (+ src1) given: the values come from there and `data` only receives the scan (no copy / add pass before it)
__global__ __launch_bounds__(256) void scan_block_kernel(uint32_t *__restrict__ data, size_t n,
                                                         uint32_t *__restrict__ block_sums,
                                                         const uint32_t *__restrict__ src0, const uint32_t *__restrict__ src1)
{
    __shared__ uint32_t wave_tot[4];
    const size_t base = (size_t)blockIdx.x * kScanPerBlock + (size_t)threadIdx.x * 8;
    uint32_t v[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        v[k] = base + k < n ? (src0 ? src0[base + k] + (src1 ? src1[base + k] : 0u) : data[base + k]) : 0u;
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

// The same for up to kScanSmallMax elements in ONE launch (a 10,000-record step is a dozen short kernels: the three
// launches of the general scan and the memset before it were 25 us of its 420): one block walks the array in pieces of
// 4,096 with a running carry.  zero[0..n_zero) is cleared on the way (the counters the next kernels add to).
constexpr size_t kScanSmallMax = 16384;   // (at 50,000 one block takes 36-49 us: more than the three launches it replaces)
__global__ __launch_bounds__(1024) void scan_small_kernel(uint32_t *__restrict__ data, uint32_t n, const uint32_t *__restrict__ src0,
                                                          const uint32_t *__restrict__ src1, uint32_t *__restrict__ zero,
                                                          uint32_t n_zero)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (tid < n_zero)
        zero[tid] = 0;
    if (tid == 0)
        carry_s = 0;
    __syncthreads();
    constexpr int PER = 4;   // elements per thread and piece: 4,096 per round of the block
    auto load_piece = [&](uint32_t at, uint32_t (&v)[PER]) {   // (n > 0; unconditional loads: a predicated one is a branch and a full wait)
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint32_t i = min(at + k, n - 1u);
            v[k] = src0 ? src0[i] + (src1 ? src1[i] : 0u) : data[i];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k)
            v[k] = at + k < n ? v[k] : 0u;
    };
    uint32_t nx[PER] = {0, 0, 0, 0};
    if (n)
        load_piece(PER * tid, nx);
    for (uint32_t base = 0; base < n; base += PER * 1024) {
        const uint32_t at = base + PER * tid;
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            v[k] = nx[k];
            sum += v[k];
        }
        if (base + PER * 1024 < n)
            load_piece(at + PER * 1024, nx);   // the next piece is on its way while this one is scanned
        uint32_t incl = sum, up;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            up = __shfl_up(incl, o);
            if (lane >= (uint32_t)o) incl += up;
        }
        if (lane == 63u)
            wave_tot[wv] = incl;
        __syncthreads();
        uint32_t off = carry_s, total = 0;
        for (uint32_t w = 0; w < 16; ++w) {
            if (w < wv) off += wave_tot[w];
            total += wave_tot[w];
        }
        uint32_t run = off + incl - sum;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (at + k < n)
                data[at + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 0)
            carry_s += total;
        __syncthreads();
    }
}

// ... and up to kScanMidMax in one launch of several blocks without a word between them: block b scans elements
// [4,096 b, 4,096 (b + 1)) and first sums everything in front of them itself (50,000 records: 13 blocks, 100 KB from L2 each
// on average) — the general scan's three launches and their gaps were 25 us of a 2.8 ms step, 30 of the shared preparation's
// 0.25 ms (it scans twice).
constexpr size_t kScanMidMax = 262144;
__global__ __launch_bounds__(1024) void scan_mid_kernel(uint32_t *__restrict__ data, uint32_t n, const uint32_t *__restrict__ src0,
                                                        const uint32_t *__restrict__ src1, uint32_t *__restrict__ zero, uint32_t n_zero)
{
    __shared__ uint32_t wave_tot[16], wave_pre[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    constexpr uint32_t PER = 4, PIECE = PER * 1024;
    if (blockIdx.x == 0 && tid < n_zero)
        zero[tid] = 0;
    auto value = [&](uint32_t i) { return src0[i] + (src1 ? src1[i] : 0u); };   // (never `data`: see below)
    // this block's piece first (its loads are in flight while the prefix is summed) ...
    const uint32_t at = blockIdx.x * PIECE + PER * tid;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k)
        v[k] = value(min(at + k, n - 1u));   // (n > 0; unconditional loads, masked below)
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        v[k] = at + k < n ? v[k] : 0u;
        sum += v[k];
    }
    // ... then everything in front of it (in place, `data` in front of this piece may already hold another block's
    // scan: the values come from src0 / src1 then, which no block writes; in place without src0 is not offered)
    uint32_t pre = 0;
    const uint32_t end = blockIdx.x * PIECE;
    for (uint32_t i = tid; i < end; i += 16u * 1024u) {   // sixteen loads (of each array) in flight: this loop is latency
        uint32_t t[16];
#pragma unroll
        for (uint32_t u = 0; u < 16; ++u)
            t[u] = value(min(i + u * 1024u, end - 1u));   // (unconditional: a predicated load is a branch and a full wait)
#pragma unroll
        for (uint32_t u = 0; u < 16; ++u)
            pre += i + u * 1024u < end ? t[u] : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        pre += __shfl_xor(pre, o);
    uint32_t incl = sum, up;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        up = __shfl_up(incl, o);
        if (lane >= (uint32_t)o) incl += up;
    }
    if (lane == 63u)
        wave_tot[wv] = incl;
    if (lane == 0)
        wave_pre[wv] = pre;
    __syncthreads();
    uint32_t off = 0;
    for (uint32_t w = 0; w < 16; ++w) {
        off += wave_pre[w];
        if (w < wv) off += wave_tot[w];
    }
    uint32_t run = off + incl - sum;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        if (at + k < n)
            data[at + k] = run;
        run += v[k];
    }
}

// totals[k] += sum of a[k][0..n) for the two count arrays of the pack (totals zeroed by the caller)
__global__ __launch_bounds__(256) void sum2_u32_kernel(const uint32_t *__restrict__ a0, const uint32_t *__restrict__ a1, size_t n,
                                                       unsigned long long *__restrict__ totals)
{
    unsigned long long s0 = 0, s1 = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        s0 += a0[i];
        s1 += a1[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o);
        s1 += __shfl_xor(s1, o);
    }
    // one atomic per block and total (per wave they were 1,600 same-address atomics at 50,000 records: 12 us)
    __shared__ unsigned long long part[4][2];
    if ((threadIdx.x & 63u) == 0) {
        part[threadIdx.x >> 6][0] = s0;
        part[threadIdx.x >> 6][1] = s1;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const unsigned long long t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (t)
            atomicAdd(&totals[threadIdx.x], t);
    }
}

// What the host wants to know after an upload, gathered into one page-locked host block by the device itself (three
// blocking device-to-host copies of 8-64 bytes cost ~20 us each: a fifth of a 10,000-record step)
// RunsArg: the run-chunk counters of the pack (RunIndex) and where the decision goes; cnt_run == NULL: not looked for
struct RunsArg {
    uint32_t *cnt_run, *run_cold, *run_hot, *index, *ids, *state;
    uint32_t max_run;   // more run records than this: stripping stays off (the correction tables would not pay / fit)
};

// The pack's list lengths summed by many blocks (one block took 45-60 us for 50,000 records: one CU's worth of memory
// latency), on the way giving every record with fewer than kRunMin run chunks their entries back — those can never be run
// records.  totals (zeroed): [0] cold entries, [1] hot entries, [2] candidates (records with >= kRunMin run chunks),
// [3] the candidates' run-chunk entries.
__global__ __launch_bounds__(256) void list_totals_kernel(uint32_t *__restrict__ cnt0, uint32_t *__restrict__ cnt1, uint32_t n,
                                                          RunsArg runs, unsigned long long *__restrict__ totals)
{
    __shared__ unsigned long long part[4][4];
    unsigned long long v[4] = {0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        uint32_t a = cnt0[i], b = cnt1[i];
        const uint32_t chunks = runs.cnt_run ? runs.cnt_run[i] : 0u;
        if (chunks) {
            const uint32_t rc = runs.run_cold[i], rh = runs.run_hot[i];
            if (chunks < kRunMin) {
                if (rc)
                    cnt0[i] = a += rc;
                if (rh)
                    cnt1[i] = b += rh;
            } else {
                v[2] += 1;
                v[3] += rc + rh;
            }
        }
        v[0] += a;
        v[1] += b;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            v[k] += __shfl_xor(v[k], o);
        if ((threadIdx.x & 63u) == 0)
            part[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const unsigned long long t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (t)
            atomicAdd(&totals[threadIdx.x], t);
    }
}

// What the host reads after an upload (one page-locked block, written by the device: no device-to-host copies), and the
// decision about the run records: with candidates and room for their tables they are numbered in record order (1,024
// records per round, a block scan of the flags); without room every candidate gets its entries back like the others.
__global__ __launch_bounds__(1024) void report_kernel(const unsigned long long *__restrict__ first_bad,
                                                      const unsigned long long *__restrict__ stats,
                                                      const unsigned long long *__restrict__ totals,
                                                      uint32_t *__restrict__ cnt0, uint32_t *__restrict__ cnt1, uint32_t n,
                                                      unsigned long long *report, RunsArg runs)
{
    __shared__ uint32_t wave_tot[16], base_s;
    __shared__ unsigned long long back[16][2];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    unsigned long long cold = totals ? totals[0] : 0ull, hot = totals ? totals[1] : 0ull;
    const unsigned long long cand = totals && runs.cnt_run ? totals[2] : 0ull, cand_entries = totals ? totals[3] : 0ull;
    const bool strip = cand != 0 && cand <= runs.max_run;
    if (cand != 0) {   // (uniform; rare: most alignments have no record with 512 sites of N in whole chunks)
        if (tid == 0)
            base_s = 0;
        unsigned long long gave[2] = {0, 0};
        __syncthreads();
        for (uint32_t i0 = 0; i0 < n; i0 += 1024) {
            const uint32_t i = i0 + tid;
            const bool is_cand = i < n && runs.cnt_run[i] >= kRunMin;
            if (is_cand && !strip) {
                const uint32_t rc = runs.run_cold[i], rh = runs.run_hot[i];
                cnt0[i] += rc;
                cnt1[i] += rh;
                gave[0] += rc;
                gave[1] += rh;
            }
            if (strip) {
                const unsigned long long m = __ballot(is_cand);
                if (lane == 0)
                    wave_tot[wv] = (uint32_t)__builtin_popcountll(m);
                __syncthreads();
                uint32_t before = base_s, total = 0;
                for (uint32_t w = 0; w < 16; ++w) {
                    if (w < wv) before += wave_tot[w];
                    total += wave_tot[w];
                }
                if (i < n) {
                    const uint32_t h = before + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                    runs.index[i] = is_cand ? h : 0xFFFFFFFFu;
                    if (is_cand)
                        runs.ids[h] = i;
                }
                __syncthreads();
                if (tid == 0)
                    base_s += total;
                __syncthreads();
            }
        }
        if (!strip) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1)
                    gave[k] += __shfl_xor(gave[k], o);
                if (lane == 0)
                    back[wv][k] = gave[k];
            }
            __syncthreads();
            for (int w = 0; w < 16; ++w) {
                cold += back[w][0];
                hot += back[w][1];
            }
        }
    }
    if (runs.state && tid == 0) {
        runs.state[0] = strip ? (uint32_t)cand : 0u;
        runs.state[1] = strip ? 1u : 0u;
    }
    if (tid == 0)
        report[0] = *first_bad;
    else if (tid <= 8)
        report[tid] = stats ? stats[tid - 1] : 0ull;
    else if (tid == 9)
        report[9] = cold;
    else if (tid == 10)
        report[10] = hot;
    else if (tid == 11)
        report[11] = strip ? cand : 0ull;           // run records (0: the lists keep every entry)
    else if (tid == 12)
        report[12] = strip ? cand_entries : 0ull;   // entries their lists lost
}

// =============================================================================================
// per-record constants A_k
// =============================================================================================
struct RunConst {   // aconst_kernel's view of the run records (all NULL: no stripping)
    uint32_t *aent;              // [kMaxWords][aent_stride] a-words of every list entry, for corr_kernel
    size_t aent_stride;
    const uint32_t *index, *mask, *known;
    uint32_t mask_words;
};

__global__ __launch_bounds__(256) void aconst_kernel(const uint32_t *__restrict__ off,
                                                     const uint32_t *__restrict__ ent,
                                                     const ConsensusLut *__restrict__ lut, int family, int wide,
                                                     int words, uint32_t n, uint32_t npad,
                                                     uint32_t *__restrict__ aconst, RunConst rc)
{
    const uint32_t lane = threadIdx.x & 63u, r = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (r >= n)
        return;
    uint32_t acc[kMaxWords] = {0, 0, 0, 0};
    // four entries per lane and step, their loads issued together: a record that is mostly N has a list of thousands
    // of entries, and one entry per step made it a chain of that many / 64 memory latencies for the whole launch
    constexpr uint32_t UNR = 4;
    const uint32_t end = off[r + 1];
    for (uint32_t i = off[r] + lane; i < end; i += 64 * UNR) {
        uint32_t e[UNR];
#pragma unroll
        for (uint32_t u = 0; u < UNR; ++u)
            e[u] = ent[min(i + 64 * u, end - 1u)];   // (i < end; unconditional loads: all in flight together)
#pragma unroll
        for (uint32_t u = 0; u < UNR; ++u) {
            if (i + 64 * u >= end)
                continue;
            const uint32_t *a = lut->a[family][wide][(e[u] >> kSiteBits) & 7u][e[u] >> kEntryShift];
#pragma unroll
            for (int w = 0; w < kMaxWords; ++w) {
                acc[w] += a[w];
                if (rc.aent && w < words)
                    rc.aent[(size_t)w * rc.aent_stride + i + 64 * u] = a[w];
            }
        }
    }
    // a run record's constant also takes back what the identity counts on its run chunks: F over them (RunIndex)
    if (rc.index && rc.index[r] != 0xFFFFFFFFu) {
        uint32_t known = 0;
        for (uint32_t k = lane; k < rc.mask_words; k += 64) {
            uint32_t m = rc.mask[(size_t)rc.index[r] * rc.mask_words + k];
            while (m) {
                const uint32_t bit = (uint32_t)__builtin_ctz(m);
                m &= m - 1;
                known += rc.known[32u * k + bit];
            }
        }
#pragma unroll
        for (int w = 0; w < kMaxWords; ++w)
            acc[w] -= known * lut->unit[family][wide][w];
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

// known reference sites of every 128-site chunk (the F of a run chunk); hot_planes given: cold sites only
__global__ __launch_bounds__(256) void run_known_kernel(const uint4 *__restrict__ ref_planes, const uint4 *__restrict__ hot_planes,
                                                        uint32_t nchunks, uint32_t *__restrict__ known)
{
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nchunks)
        return;
    const uint4 A = ref_planes[c], G = ref_planes[nchunks + c], C = ref_planes[2 * (size_t)nchunks + c], T = ref_planes[3 * (size_t)nchunks + c];
    uint4 k;
    k.x = (A.x ^ G.x ^ C.x ^ T.x) & ~((A.x & G.x) | (C.x & T.x));
    k.y = (A.y ^ G.y ^ C.y ^ T.y) & ~((A.y & G.y) | (C.y & T.y));
    k.z = (A.z ^ G.z ^ C.z ^ T.z) & ~((A.z & G.z) | (C.z & T.z));
    k.w = (A.w ^ G.w ^ C.w ^ T.w) & ~((A.w & G.w) | (C.w & T.w));
    if (hot_planes) {
        const uint4 h = hot_planes[c];
        k.x &= ~h.x;
        k.y &= ~h.y;
        k.z &= ~h.z;
        k.w &= ~h.w;
    }
    known[c] = popc4(k);
}

// the run chunks of every run record as a bit mask, from the flags the pack left in the slots
__global__ __launch_bounds__(256) void run_masks_kernel(const uint4 *__restrict__ slots, const uint32_t *__restrict__ ids, uint32_t n_run,
                                                        uint32_t nchunks, uint32_t npad, uint32_t mask_words, uint32_t *__restrict__ mask,
                                                        uint32_t *__restrict__ mask_t)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_run * mask_words)
        return;
    const uint32_t h = i / mask_words, k = i - h * mask_words, r = ids[h];
    uint32_t m = 0;
    for (uint32_t bit = 0; bit < 32; ++bit) {
        const uint32_t c = 32u * k + bit;
        if (c < nchunks && ((slots[(size_t)c * npad + r].x >> 8) & 1u))
            m |= 1u << bit;
    }
    mask[i] = m;
    mask_t[(size_t)k * n_run + h] = m;   // [word][run record]: what the table kernel's lanes read side by side
}

// first run record of every column panel: panel_first[p] = run records with id < p * kPanelCols (ids ascending)
__global__ __launch_bounds__(256) void run_panels_kernel(const uint32_t *__restrict__ ids, uint32_t n_run, uint32_t n_panels,
                                                         uint32_t *__restrict__ panel_first)
{
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p > n_panels)
        return;
    const uint32_t key = p * kPanelCols;
    uint32_t lo = 0, hi = n_run;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (ids[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    panel_first[p] = lo;
}

// X's terms (RunIndex) for every (run record h, record r):
//     C(h, r) = - A_r(M_h)  [r's list entries in h's run chunks]  +  F(M_h n M_r) if r is a run record with a higher id
// — in the table the run ROWS read (corr[h][r]: what row h adds to column r) but not in the one the run COLUMNS read
// (corr_t[r][h]): a pair (q, t) gets corr_t[q][t's number] when t is a run record and corr[q's number][t] when q is one,
// so both A terms once and the F term once.
// A_r(M_h) = sum over chunks c of [c in M_h] x S(r, c), S(r, c) = the a-words of r's entries in chunk c: a 0/1 matrix
// (run records x chunks) times a matrix of per-chunk sums (chunks x records).  That IS a matrix product, of integers,
// and at 4,000 run records x 20,000 records x 235 chunks the VALU forms of it (a test per list entry and run record:
// 3.3 ms; LDS atomics per incidence: 1.6 ms) cost more than the pair kernel — so it goes through the matrix cores:
// v_mfma_i32_32x32x32_i8, the 32-bit sums cut into five 7-bit pieces (exact: a piece is below 128, 256 chunks of them
// stay far inside an int32; the pieces' results are shifted back together modulo 2^32, like the packed tallies).
constexpr int kSumPieces = 5;   // 7 + 7 + 7 + 7 + 4 bits of a 32-bit sum

// S7[w][piece][record][Kpad]: the per-chunk sums of every record's a-words, in 7-bit pieces (bytes); Kpad = 32 mask_words
// fold (run_index != NULL): a run record's sums also lose known[c] x unit on its own run chunks — then the product with
// another run record's mask carries the F term of the two (+F after the sign flip).  The table the run ROWS read is built
// from the folded sums, the table the run COLUMNS read from the plain ones: a pair of two run records meets the F term once.
template <int W>
__global__ __launch_bounds__(256) void chunk_sums_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ ent,
                                                         const uint32_t *__restrict__ aent, size_t aent_stride, uint32_t n,
                                                         uint32_t kpad, uint8_t *__restrict__ s7,
                                                         const uint32_t *__restrict__ run_index, const uint32_t *__restrict__ mask,
                                                         const uint32_t *__restrict__ known, const ConsensusLut *__restrict__ lut,
                                                         int family, int wide)
{
    extern __shared__ uint32_t srow[];   // [4 waves][W][kpad]
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, r = blockIdx.x * 4u + wv;
    const size_t n_tiles = (n + 31u) / 32u;
    uint32_t *mine = srow + (size_t)wv * W * kpad;
    for (uint32_t c = lane; c < W * kpad; c += 64)
        mine[c] = 0;
    if (r >= n)
        return;
    const uint32_t end = off[r + 1];
    for (uint32_t i = off[r] + lane; i < end; i += 64) {
        const uint32_t c = (ent[i] & kSiteMask) >> 7;
#pragma unroll
        for (int w = 0; w < W; ++w)
            atomicAdd(&mine[w * kpad + c], aent[(size_t)w * aent_stride + i]);
    }
    if (run_index && run_index[r] != 0xFFFFFFFFu) {
        const uint32_t *m = mask + (size_t)run_index[r] * (kpad / 32u);
        for (uint32_t c = lane; c < kpad; c += 64)
            if ((m[c >> 5] >> (c & 31u)) & 1u) {
#pragma unroll
                for (int w = 0; w < W; ++w)
                    mine[w * kpad + c] -= known[c] * lut->unit[family][wide][w];
            }
    }
    // (one wave writes and reads its own LDS rows: in order)
    for (uint32_t c4 = lane * 4; c4 < kpad; c4 += 256) {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint32_t v[4] = {mine[w * kpad + c4], mine[w * kpad + c4 + 1], mine[w * kpad + c4 + 2], mine[w * kpad + c4 + 3]};
#pragma unroll
            for (int p = 0; p < kSumPieces; ++p) {
                const uint32_t sh = 7u * p;
                const uint32_t packed = ((v[0] >> sh) & 127u) | ((v[1] >> sh) & 127u) << 8 | ((v[2] >> sh) & 127u) << 16 |
                                        ((v[3] >> sh) & 127u) << 24;
                // laid out as corr_mfma_kernel's lanes read it: per (tile of 32 records, step of 32 chunks) one contiguous KB,
                // lane (half, record in tile) at 16 bytes x (32 half + record): a wave's operand load is eight whole lines
                const size_t tile = r >> 5, ks = c4 >> 5, hf = (c4 >> 4) & 1u;
                *reinterpret_cast<uint32_t *>(s7 + ((((size_t)w * kSumPieces + p) * n_tiles + tile) * (kpad / 32u) + ks) * 1024u +
                                              (hf * 32u + (r & 31u)) * 16u + (c4 & 15u)) = packed;
            }
        }
    }
}

typedef int v4i32_t __attribute__((ext_vector_type(4)));
typedef int v16i32_t __attribute__((ext_vector_type(16)));

// 16 mask bits -> 16 bytes of 0 / 1 (the A operand's fragment: a lane's 16 consecutive k)
__device__ __forceinline__ v4i32_t spread16(uint32_t bits)
{
    v4i32_t o;
    o.x = (int)(((bits & 15u) * 0x00204081u) & 0x01010101u);
    o.y = (int)((((bits >> 4) & 15u) * 0x00204081u) & 0x01010101u);
    o.z = (int)((((bits >> 8) & 15u) * 0x00204081u) & 0x01010101u);
    o.w = (int)((((bits >> 12) & 15u) * 0x00204081u) & 0x01010101u);
    return o;
}

// One block = 128 run records x 128 records: each of the 4 waves 32 records against four 32-run-record sub-tiles.  A = the run records' masks as 0/1
// bytes, B = one 7-bit piece of the records' per-chunk sums; lane l holds row / column l & 31 and the 16 consecutive
// k of half l >> 5 of every 32-chunk step, in both operands (the same k in the same place is all the sum needs).
// D (dtype-independent on gfx950): column = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
// TRANSPOSED: the result goes to corr_t[w][record][run record] (through LDS: a row of 32 run records per record), else to
// corr[w][run record][record] (32 consecutive records per lane group as the accumulators lie).
template <int W, bool TRANSPOSED>
__global__ __launch_bounds__(256) void corr_mfma_kernel(const uint32_t *__restrict__ mask_t, uint32_t mask_words,
                                                        const uint8_t *__restrict__ s7, uint32_t n, uint32_t n_run,
                                                        uint32_t *__restrict__ table)
{
    // a wave = 32 records x kRunTiles x 32 run records: the B fragment (16 bytes of a record's sums per lane, rows 32 k
    // apart: half a cache line of use per line fetched) is loaded once and meets kRunTiles mask fragments
    constexpr int kRunTiles = 4;
    __shared__ uint32_t turn[TRANSPOSED ? 4 : 1][TRANSPOSED ? 32 : 1][33];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, half = lane >> 5, idx = lane & 31u;
    const uint32_t r_col = blockIdx.x * 128u + wv * 32u + idx;     // this lane's record (B's column)
    const uint32_t h_base = blockIdx.y * (32u * kRunTiles);
    const bool b_live = r_col < n;
    // masks transposed ([k-step][run record]: 32 lanes read 128 contiguous bytes), sums tiled (see chunk_sums_kernel)
    const uint32_t *mcol[kRunTiles];
    bool a_live[kRunTiles];
#pragma unroll
    for (int t = 0; t < kRunTiles; ++t) {
        const uint32_t h_row = h_base + 32u * t + idx;             // this lane's run record of sub-tile t (A's row)
        a_live[t] = h_row < n_run;
        mcol[t] = mask_t + min(h_row, n_run - 1u);
    }
    const size_t n_tiles = (n + 31u) / 32u, r_tile = blockIdx.x * 4u + wv;
#pragma unroll 1
    for (int w = 0; w < W; ++w) {
        uint32_t out[kRunTiles][16];
#pragma unroll
        for (int t = 0; t < kRunTiles; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v)
                out[t][v] = 0;
#pragma unroll 1
        for (int p = 0; p < kSumPieces; ++p) {
            const uint8_t *brow = s7 + (((size_t)w * kSumPieces + p) * n_tiles + min<size_t>(r_tile, n_tiles - 1)) * mask_words * 1024u +
                                  lane * 16u;
            v16i32_t acc[kRunTiles];
#pragma unroll
            for (int t = 0; t < kRunTiles; ++t)
                acc[t] = v16i32_t{};
            // the operands of step ks + 1 are loaded before the MFMAs of step ks are issued: with four 16-register
            // accumulators a SIMD holds two or three of these waves, too few to hide a load per step behind each other
            // Eight k-steps (256 chunks) at a time, ALL their operands loaded before the first MFMA: a step is ~100 ns of
            // issue and a load ~1.5 us away at two waves per SIMD, so operands fetched one step ahead (the first form of
            // this loop) left the wave waiting for memory forty times per piece: 0.5 ms per table where the MFMAs are
            // worth 0.05.  Every load is unconditional, from a clamped address, its value dropped by a select afterwards
            // (written as conditional loads each became a branch with its own wait).
            constexpr uint32_t KB = 8;
            for (uint32_t k0 = 0; k0 < mask_words; k0 += KB) {
                uint4 bv[KB];
                uint32_t word[KB][kRunTiles];
#pragma unroll
                for (uint32_t j = 0; j < KB; ++j) {
                    const uint32_t kc = min(k0 + j, mask_words - 1u);
                    bv[j] = *reinterpret_cast<const uint4 *>(brow + 1024u * kc);
#pragma unroll
                    for (int t = 0; t < kRunTiles; ++t)
                        word[j][t] = mcol[t][(size_t)kc * n_run];
                }
#pragma unroll
                for (uint32_t j = 0; j < KB; ++j) {
                    const bool on = k0 + j < mask_words;
                    v4i32_t bf;
                    bf.x = (b_live && on) ? (int)bv[j].x : 0;
                    bf.y = (b_live && on) ? (int)bv[j].y : 0;
                    bf.z = (b_live && on) ? (int)bv[j].z : 0;
                    bf.w = (b_live && on) ? (int)bv[j].w : 0;
#pragma unroll
                    for (int t = 0; t < kRunTiles; ++t) {
                        const uint32_t bits = (a_live[t] && on) ? (word[j][t] >> (16u * half)) & 0xFFFFu : 0u;
                        acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(spread16(bits), bf, acc[t], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < kRunTiles; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    out[t][v] += (uint32_t)acc[t][v] << (7u * p);
        }
#pragma unroll
        for (int t = 0; t < kRunTiles; ++t) {
            const uint32_t h0 = h_base + 32u * t;
            if (h0 >= n_run)   // (uniform)
                continue;
            if constexpr (!TRANSPOSED) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const uint32_t h = h0 + (v & 3u) + 8u * (v >> 2) + 4u * half;
                    if (b_live && h < n_run)
                        table[((size_t)w * n_run + h) * n + r_col] = 0u - out[t][v];
                }
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    turn[wv][idx][(v & 3u) + 8u * (v >> 2) + 4u * half] = 0u - out[t][v];   // [record][run record]
                // (a wave's own tile: LDS operations of one wave complete in order)
                for (uint32_t e = lane; e < 32u * 32u; e += 64) {
                    const uint32_t rr = e >> 5, hh = e & 31u;
                    const uint32_t r = blockIdx.x * 128u + wv * 32u + rr, h = h0 + hh;
                    if (r < n && h < n_run)
                        table[((size_t)w * n + r) * n_run + h] = turn[wv][rr][hh];
                }
            }
        }
    }
}

// =============================================================================================
// pair kernel
// =============================================================================================
template <int FAM, bool WIDE>
struct Pack {
    static constexpr int NT = FAM == FAM_NHIGH ? 1 : FAM == FAM_RAW ? 2 : FAM == FAM_K80 ? 3 : 4;
    static constexpr int W = WIDE ? NT : (NT + 1) / 2;
    // tallies of the hot columns, written by the dense kernels as DST_OUT_TALLY16 (narrow) / DST_OUT_TALLY (wide):
    // the 16-bit layout of 2 and 4 tallies IS the packed accumulator word
    static __device__ __forceinline__ void add_hot(uint32_t *t, const void *hot, uint64_t at)
    {
        if constexpr (WIDE) {
#pragma unroll
            for (int k = 0; k < NT; ++k)
                t[k] += static_cast<const uint32_t *>(hot)[at * NT + k];
        } else if constexpr (NT == 1) {
            t[0] += static_cast<const uint16_t *>(hot)[at];
        } else if constexpr (NT == 2) {
            t[0] += static_cast<const uint32_t *>(hot)[at];
        } else if constexpr (NT == 3) {
            const uint16_t *p = static_cast<const uint16_t *>(hot) + at * 3;
            t[0] += (uint32_t)p[0] | (uint32_t)p[1] << 16;
            t[1] += p[2];
        } else {
            const uint2 v = static_cast<const uint2 *>(hot)[at];
            t[0] += v.x;
            t[1] += v.y;
        }
    }
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

// the run records' corrections as the pair kernel reads them (RunIndex; ids == NULL: none).  Square launches only.
struct RunPair {
    const uint32_t *index, *ids, *panel_first, *corr, *corr_t;
    uint32_t n_run, n;
};

// which instantiations write their results in address-aligned quarters (see ALIGNED in the kernel): the single-word
// families whose time is the result stream itself; jc69's output phase is bound by its f64 logarithm instead
template <int FAM, bool WIDE, int OUT>
constexpr bool aligned_output()
{
#ifdef DST_DBG_OLDMAP
    return false;
#elif defined(DST_DBG_ALIGN_JC69)
    return Pack<FAM, WIDE>::W == 1;
#else
    return Pack<FAM, WIDE>::W == 1 && OUT != DST_JC69;
#endif
}

constexpr int kBlockWaves = 8;   // waves of a workgroup: event waves + output waves

// Event waves per workgroup; the others are output waves.  The events of a low-diversity batch are few (two rows x
// ~65 entries) and their loads are pipelined, so two waves carry them; the finalisation is what needs the issue
// slots — more so the more f64 work a result takes (50,000 x 30,000, 4+4 -> 2+6 -> 1+7 waves: raw 2.40 -> see
// profiles/r02; tn93 16.7 ms at 4+4).
template <int FAM, bool WIDE, int OUT>
constexpr int event_waves()
{
#ifdef DST_DBG_EVWAVES
    return DST_DBG_EVWAVES;
#else
    // tn93 is bound by the f64 finalisation, and a wave that only gathers events leaves its SIMD with one finalising wave
    // where the others have two: all 8 waves in both roles instead (kBlockWaves: no event role at all) — 15.0 -> 13.5 ms
    // at 50,000 x 30,000; k80 gains 1 % that way and jc69 loses 9 %, they keep their roles (k80 with two event waves:
    // with one, 5.16 ms, the batch's ~130 entries took three slices and the seven output waves waited for them: 4.61)
    return OUT == DST_TN93 ? (WIDE ? 1 : kBlockWaves) : 2;
#endif
}
// 32-bit words of dynamic LDS before the logarithm table: accumulators, h table, row offsets, (ALIGNED) A(column)
template <int FAM, bool WIDE, int OUT, int EW>
constexpr size_t cpair_smem_words()
{
    constexpr int W = Pack<FAM, WIDE>::W;
#ifdef DST_DBG_RB
    constexpr int RBL = DST_DBG_RB;
#else
    constexpr int RBL = kAccRows;
#endif
    size_t words = (size_t)(EW == kBlockWaves ? 1 : 2) * RBL * W * kPanelCols + (size_t)kRefClasses * 256 * W + kTileRowsMax + 1 +
                   (aligned_output<FAM, WIDE, OUT>() ? kPanelCols : 0);
    return (words + 3) & ~(size_t)3;   // the table's entries are 16 bytes
}
// Launches with many events per pair or long lists (ConsensusLaunch::heavy_events == 1): 4 event + 4 output waves.
constexpr int kEventWavesHeavy = 4;
// More than one event per pair (heavy_events == 2): such launches are bound by the event side's chains of dependent
// loads, and what hides those is waves.  EW == kBlockWaves is the variant for them: no roles — all 8 waves apply a
// batch's events, then all 8 write it out — so one accumulator buffer is enough (29 KB of LDS instead of 45 for raw:
// more blocks per CU) and every wave of every block is an event wave (20,000 x 30,000 with 5 % of the records half N:
// 17.6 -> 12.9 ms; below one event per pair the lost overlap of the two sides costs more than it gives).
#ifdef DST_DBG_HEAVY_EW
constexpr int kEventWavesAll = DST_DBG_HEAVY_EW;
#else
constexpr int kEventWavesAll = kBlockWaves;
#endif

// the value must be in its register HERE (an empty asm the compiler cannot move a definition across)
__device__ __forceinline__ void pin(uint32_t &v) { asm volatile("" : "+v"(v)); }

// Two adjacent results, 8-byte aligned: one 16-byte store — NONTEMPORAL: the results are written once and never read
// by this kernel; left to the default policy the 10 GB result stream of a 50,000-record run evicts the lookup tables
// the event waves gather from out of the L2 (raw, 50,000 x 30,000: 3.2 ms with plain stores, 2.5 ms with these).
typedef double F64x2 __attribute__((ext_vector_type(2), aligned(8)));
typedef long long I64x2 __attribute__((ext_vector_type(2), aligned(8)));
template <typename T>
__device__ __forceinline__ void store_result(T *p, T v)
{
#ifdef DST_DBG_PLAIN_STORES
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}
__device__ __forceinline__ void store_result2(double *p, double a, double b)
{
    F64x2 v = {a, b};
#ifdef DST_DBG_PLAIN_STORES
    *reinterpret_cast<F64x2 *>(p) = v;
#else
    __builtin_nontemporal_store(v, reinterpret_cast<F64x2 *>(p));
#endif
}
__device__ __forceinline__ void store_result2(int64_t *p, int64_t a, int64_t b)
{
    I64x2 v = {(long long)a, (long long)b};
#ifdef DST_DBG_PLAIN_STORES
    *reinterpret_cast<I64x2 *>(p) = v;
#else
    __builtin_nontemporal_store(v, reinterpret_cast<I64x2 *>(p));
#endif
}

// One block = rows [i0, i1) x one panel of up to kPanelCols column records; the rows go through kAccRows at a
// time (a "batch"; their lists are adjacent in the CSR, so a batch is one run of entries).  512 threads in two
// roles that work on consecutive batches at the same time, with the accumulators double-buffered in LDS:
//   event waves (EW): A) lane k takes entry k of the batch's lists: (site, reference class, nibble) -> the 32-byte
//      lookup-table entry of (panel, site), which IS the bucket: its size and up to 15 column entries (record in the
//      panel | nibble), or 13 and the place of the others;  B) every candidate event (column record, nibble) gets
//      h_k from the table in LDS and goes into the (row, column) accumulators with ds_add_u32.  What a larger bucket
//      holds beyond the inline entries is shared out over the wave (scan of the sizes + search by shuffles); batches
//      with more entries than the event lanes run extra slices.  The two dependent loads (entry, table entry) are
//      software-pipelined in registers across batches.  No barrier inside A/B.
//   output waves (8 - EW): C) for each row of the PREVIOUS batch every thread walks its column PAIRS (16-byte
//      nontemporal stores): accumulator + A(column) + A(row) + F, unpack, finalise, store in canonical order; touched
//      accumulators are reset on the way.
// One barrier per batch.  Why two roles: gfx950 counts loads and stores in ONE in-order counter (vmcnt), so a
// wave that has just issued its result stores cannot consume a younger load before those stores have
// landed in HBM; the event waves' chains of dependent random loads never queue behind a store this way.
// Why nontemporal stores: left to the default policy the 10 GB result stream of a 50,000-record launch flows through
// the L2 and evicts the panel's lookup table the event waves gather from (FETCH_SIZE 0.88 GB -> 0.15 GB per launch,
// raw 3.2 -> 2.5 ms; profiles/r02).
template <int FAM, bool WIDE, int OUT, int EW>
__global__ __launch_bounds__(64 * kBlockWaves, OUT == DST_TN93 ? 4 : 2) void consensus_pair_kernel(
    const uint32_t *__restrict__ row_off, const uint32_t *__restrict__ row_ent,
    const uint32_t *__restrict__ row_a, uint32_t row_npad, const uint4 *__restrict__ site_inl,
    const uint32_t *__restrict__ site_ent, const uint32_t *__restrict__ col_a, uint32_t col_npad,
    uint32_t n_sites, const ConsensusLut *__restrict__ lut, FWords fw,
    const ConsensusTile *__restrict__ tiles, void *__restrict__ out_v, const uint32_t *__restrict__ q_counts,
    const uint32_t *__restrict__ t_counts, uint32_t n_cols, uint32_t row_begin, uint64_t out_base, int square,
    const void *__restrict__ hot, RunPair rp)
{
    using P = Pack<FAM, WIDE>;
    constexpr int W = P::W, NT = P::NT;
#ifdef DST_DBG_RB
    constexpr int RB = DST_DBG_RB;
#else
    constexpr int RB = kAccRows;
#endif
    constexpr bool UNI = EW == kBlockWaves;            // every wave in both roles, one after the other (event-heavy launches)
    constexpr int NOW = UNI ? kBlockWaves : kBlockWaves - EW;   // EW event waves, NOW output waves
    constexpr uint32_t kEventLanes = 64 * EW;          // entries of a batch the register pipeline carries
    constexpr uint32_t OT = 64 * NOW;                  // output threads
    constexpr int PAIRS = (kPanelCols + 2 * OT - 1) / (2 * OT);   // column pairs per output thread (panel-relative mapping)
    constexpr uint32_t ACC = RB * W * kPanelCols;  // words of one accumulator buffer
    extern __shared__ uint32_t smem[];
    uint32_t *acc = smem;                                  // [2 (UNI: 1)][RB][W][kPanelCols]
    constexpr uint32_t NBUF = UNI ? 1 : 2;
    uint32_t *hlut = smem + NBUF * ACC;                    // [kRefClasses][16][16][W]: h_k of this family
    uint32_t *rofs = hlut + kRefClasses * 256 * W;         // [kTileRowsMax + 1] list offsets of the tile's rows
    // Single-word families (n, n_high, raw, jc69 below 65,536 sites) are bound by WRITING the results, and the
    // write rate depends on the store pattern (tools/ubench/store_rate.hip on MI355X, 10 GB triangle: every output
    // wave storing 1 KB pieces 4 KB apart from an 8-byte-aligned row start 3.5-3.9 TB/s; each wave one contiguous
    // quarter of the row cut at 128-byte lines of the absolute address 4.7-5.1 TB/s = the rate of a plain aligned
    // fill).  So for them (ALIGNED) a thread's columns follow the ADDRESS of the row, not the panel, and A(column)
    // comes from an LDS copy of the panel's constants instead of registers.
    constexpr bool ALIGNED = aligned_output<FAM, WIDE, OUT>();
    uint32_t *cola = rofs + kTileRowsMax + 1;              // ALIGNED: [kPanelCols] A(column) of this panel
    // the logarithm's table (2 KB) in LDS for the measures that take one: an output wave must not load from global
    // memory between its result stores (one in-order counter for loads and stores: the load waits for every earlier store)
    constexpr bool LOGS = OUT == DST_JC69 || OUT == DST_K80 || OUT == DST_TN93;
    LogEntry *logtab = reinterpret_cast<LogEntry *>(smem + cpair_smem_words<FAM, WIDE, OUT, EW>());

    const ConsensusTile tile = tiles[blockIdx.x];
    const uint32_t panel0 = tile.panel * kPanelCols;
    const uint32_t pcols = min(kPanelCols, n_cols - panel0);
    const bool event_role = !UNI && threadIdx.x >= OT;
    const uint32_t tid = event_role ? threadIdx.x - OT : threadIdx.x, lane = threadIdx.x & 63u;   // index within the role
    const uint32_t trows = tile.i1 - tile.i0;              // <= kTileRowsMax
    const uint32_t nbatch = (trows + RB - 1) / RB;
    for (uint32_t k = threadIdx.x; k < NBUF * ACC; k += blockDim.x)
        acc[k] = 0;
    for (uint32_t k = threadIdx.x; k < kRefClasses * 256 * W; k += blockDim.x)
        hlut[k] = (&lut->h[FAM][WIDE ? 1 : 0][0][0][0][0])[(k / W) * kMaxWords + k % W];
    if (threadIdx.x <= trows)
        rofs[threadIdx.x] = row_off[tile.i0 + threadIdx.x];
    if constexpr (ALIGNED)
        for (uint32_t k = threadIdx.x; k < kPanelCols; k += blockDim.x) {
            const uint32_t av = col_a[panel0 + min(k, pcols - 1u)];   // (unconditional, masked: see HOIST below)
            cola[k] = k < pcols ? av : 0u;
        }
    if constexpr (LOGS)
        if (threadIdx.x < 128)
            logtab[threadIdx.x] = kLogTab[threadIdx.x];
    // A(column) of an output thread's column pairs — and, for tn93, the columns' base counts — are constant over the
    // rows of the tile: kept in registers for the whole tile (the panel-relative mapping).  Loading them inside the
    // output loop would put a global load between the result stores, and gfx950 makes that load wait for every store
    // issued before it (tn93 with the counts re-read per pair: 15.6 ms; with the load removed for a measurement 14.3).
    constexpr bool HOIST = !ALIGNED;
    constexpr bool HOIST_TC = HOIST && OUT == DST_TN93;
    constexpr int TCW = WIDE ? 4 : 2;   // below 65,536 sites a count fits 16 bits: two words per column
    uint32_t ca[HOIST ? PAIRS : 1][2][W];
    uint32_t tcp[HOIST_TC ? PAIRS : 1][2][TCW];
    if constexpr (HOIST) {
#pragma unroll
        for (int j = 0; j < PAIRS; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t k = 2 * tid + 2 * OT * j + h;
                const bool in = !event_role && k < pcols;
                // (loads from a clamped column, masked afterwards: a predicated load compiles to a branch and a full
                // wait, and these PAIRS x 2 x W loads of a tile's prologue then come one memory latency after another)
                const uint32_t kc = min(k, pcols - 1u);
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const uint32_t av = col_a[(size_t)w * col_npad + panel0 + kc];
                    ca[j][h][w] = in ? av : 0u;
                    // keep the VALUE in a register: left to itself hipcc re-loads it inside the output loop
                    asm volatile("" : "+v"(ca[j][h][w]));
                }
                if constexpr (HOIST_TC) {
                    const uint4 tv = reinterpret_cast<const uint4 *>(t_counts)[panel0 + kc];
                    const uint4 t4 = in ? tv : make_uint4(0, 0, 0, 0);
                    if constexpr (WIDE) {
                        tcp[j][h][0] = t4.x, tcp[j][h][1] = t4.y, tcp[j][h][2] = t4.z, tcp[j][h][3] = t4.w;
                    } else {
                        tcp[j][h][0] = t4.x | t4.y << 16;
                        tcp[j][h][1] = t4.z | t4.w << 16;
                    }
#pragma unroll
                    for (int x = 0; x < TCW; ++x)
                        asm volatile("" : "+v"(tcp[j][h][x]));
                }
            }
    }
    __syncthreads();

    // ---- event waves: the three stages of A, pipelined in registers across batches (entry of batch b+3,
    // bucket of batch b+2, bucket entries of batch b+1 are in flight while batch b's events are applied)
    struct Entry {   // stage 1: this lane's entry of a batch
        uint32_t e, rb;
        bool valid;
    };
    struct Inl {     // stage 2: the 32-byte lookup-table entry of its site in this panel = the bucket itself
        uint4 lo, hi;        // halfword 0: entries in the bucket (saturating); halfwords 1..15: record-in-panel | nibble << 11
        uint32_t meta;
    };
    auto load_entry = [&](uint32_t b, uint32_t first) {   // entry `first + tid` of batch b's run
        Entry en{0u, 0u, false};
        if (b < nbatch) {
            const uint32_t r0 = b * RB;
            const uint32_t ei = rofs[r0] + first + tid;
            if (ei < rofs[min(r0 + RB, trows)]) {
                en.valid = true;
                en.e = row_ent[ei];
#pragma unroll
                for (int r = 1; r < RB; ++r)
                    en.rb += ei >= rofs[min(r0 + r, trows)];
            }
        }
        return en;
    };
    auto load_inl = [&](const Entry &en) {
        Inl t{make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), 0u};
        if (en.valid) {
            const uint4 *p = site_inl + 2 * ((size_t)tile.panel * n_sites + (en.e & kSiteMask));
            t.lo = p[0];
            t.hi = p[1];
            t.meta = en.rb << 8 | ((en.e >> kSiteBits) & 7u) << 4 | (en.e >> kEntryShift);
        }
        return t;
    };
    // B for one lane's bucket of batch b: up to kInlineEvents events straight from the table entry; what a larger
    // bucket holds beyond them comes from the bucket array, shared out over the wave
    auto apply_bucket = [&](const Inl &t, uint32_t b) {
        const uint32_t q0 = tile.i0 + b * RB;
        uint32_t *bacc = acc + (UNI ? 0u : b & 1u) * ACC;
        // one candidate event: column record + nibble from the bucket, h_k from the table, into the accumulators.
        // meta = row of the batch << 8 | (reference class << 4 | row nibble): the table row and the accumulator
        // row are per-entry values; per event there is the column's nibble and its place in the panel.
        // Only tiles on the diagonal have to test t > q.
        const bool diag = square && panel0 <= tile.i1;
        auto apply = [&](uint32_t col, uint32_t nib, uint32_t meta) {
            const uint32_t rb = meta >> 8;
            if (diag && panel0 + col <= q0 + rb)
                return;
            const uint32_t *h = hlut + (((meta & 255u) << 4) | nib) * W;
            uint32_t *a = bacc + rb * W * kPanelCols + col;
#pragma unroll
            for (int w = 0; w < W; ++w)
                atomicAdd(&a[w * kPanelCols], h[w]);   // adding 0 is cheaper than testing for it
        };
        const uint32_t cnt = t.lo.x & 0xFFFFu;
        const uint32_t w8[8] = {t.lo.x, t.lo.y, t.lo.z, t.lo.w, t.hi.x, t.hi.y, t.hi.z, t.hi.w};
        const bool over = cnt > kInlineEvents;   // then the last word is the place of the other entries, not two of them
        const uint32_t inl_n = over ? kInlineOverflowing : cnt;
#pragma unroll
        for (uint32_t k = 1; k <= kInlineEvents; ++k)
            if (k <= inl_n) {
                const uint32_t e16 = (k & 1u) ? w8[k >> 1] >> 16 : w8[k >> 1] & 0xFFFFu;
                apply(e16 & (kPanelCols - 1u), e16 >> 11, t.meta);
            }
        if (__ballot(over)) {
            const uint32_t o0 = w8[7], ex = over ? cnt - kInlineOverflowing : 0u;
            uint32_t incl = ex, up;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                up = __shfl_up(incl, o);
                if (lane >= (uint32_t)o) incl += up;
            }
            const uint32_t total = __shfl(incl, 63), start = incl - ex;
            // entry i of the wave's overflow entries: whose bucket (search by shuffles), then its load; the loads of
            // the NEXT 64 are issued before this 64 are applied (event-heavy launches spend their time here)
            auto fetch = [&](uint32_t i0, uint32_t &c, uint32_t &meta) {
                const uint32_t i = i0 + lane;
                uint32_t lo = 0;
#pragma unroll
                for (int st = 32; st > 0; st >>= 1) {
                    const uint32_t v = __shfl(start, (int)(lo + st));
                    if (v <= i) lo += st;   // the last lane whose range starts at or before i (sizes may be 0)
                }
                const uint32_t s_o0 = __shfl(o0, (int)lo), s_start = __shfl(start, (int)lo);
                meta = __shfl(t.meta, (int)lo);
                c = i < total ? site_ent[s_o0 + (i - s_start)] : 0u;
            };
            // kOvf x 64 entries per round, the next round's loads issued before this round's events are applied
            // (event-heavy launches live in this loop; deeper than 2 costs more in registers and idle searches than it hides)
            constexpr uint32_t kOvf = DST_OVF_DEPTH;
            uint32_t c_nx[kOvf], m_nx[kOvf];
#pragma unroll
            for (uint32_t u = 0; u < kOvf; ++u)
                fetch(64 * u, c_nx[u], m_nx[u]);
            for (uint32_t i0 = 0; i0 < total; i0 += 64 * kOvf) {
                uint32_t c[kOvf], meta[kOvf];
#pragma unroll
                for (uint32_t u = 0; u < kOvf; ++u) {
                    c[u] = c_nx[u];
                    meta[u] = m_nx[u];
                }
                if (i0 + 64 * kOvf < total) {
#pragma unroll
                    for (uint32_t u = 0; u < kOvf; ++u)
                        fetch(i0 + 64 * (kOvf + u), c_nx[u], m_nx[u]);
                }
#pragma unroll
                for (uint32_t u = 0; u < kOvf; ++u)
                    if (i0 + 64 * u + lane < total)
                        apply(c[u] & (kPanelCols - 1u), c[u] >> kEntryShift, meta[u]);
            }
        }
    };
    // ---- the run records' corrections of batch b (RunIndex): what the run COLUMNS of this panel add to each row of the
    // batch (corr_t[row][h0 .. h1): contiguous), and, for a row that is a run record itself, what it adds to every
    // column of the panel (corr[h][panel0 ..): contiguous).  Plain adds into the accumulators, like events.
    const uint32_t run_h0 = rp.ids ? rp.panel_first[tile.panel] : 0u, run_h1 = rp.ids ? rp.panel_first[tile.panel + 1] : 0u;
    auto apply_runs = [&](uint32_t b, uint32_t t, uint32_t nt) {   // thread t of nt
        if (!rp.ids)
            return;
        const uint32_t q0 = tile.i0 + b * RB, nrows = min((uint32_t)RB, tile.i1 - q0);
        uint32_t *bacc = acc + (UNI ? 0u : b & 1u) * ACC;
        // (four columns per lane and round, and every row of the batch, loaded before the first value is added: the values
        // are independent, only the adds are ordered — a row at a time this was one memory latency per row on the event
        // waves.  Loads from clamped indices, masked afterwards: a predicated load is a branch and a full wait.  Measured
        // and not kept: the next batch's first round fetched across the batch barrier (3.53 -> 3.68 ms at 50,000 records
        // with 5 % run records), ten columns per lane and round for the run rows (-> 3.79).)
        constexpr uint32_t UNR = 4;
        for (uint32_t h0 = run_h0 + t; h0 < run_h1; h0 += nt * UNR) {
            uint32_t col[UNR], v[RB][UNR][W];
#pragma unroll
            for (uint32_t u = 0; u < UNR; ++u) {
                const uint32_t h = h0 + nt * u, hc = min(h, run_h1 - 1u);
                const uint32_t id = rp.ids[hc];
                col[u] = h < run_h1 ? id - panel0 : 0xFFFFFFFFu;
#pragma unroll
                for (uint32_t rb = 0; rb < (uint32_t)RB; ++rb) {
                    const uint32_t q = min(q0 + rb, tile.i1 - 1u);
#pragma unroll
                    for (int w = 0; w < W; ++w)
                        v[rb][u][w] = rp.corr_t[((size_t)w * rp.n + q) * rp.n_run + hc];
                }
            }
#pragma unroll
            for (uint32_t rb = 0; rb < (uint32_t)RB; ++rb) {
                if (rb >= nrows)
                    break;
                uint32_t *racc = bacc + rb * W * kPanelCols;
#pragma unroll
                for (uint32_t u = 0; u < UNR; ++u) {
                    if (col[u] == 0xFFFFFFFFu || (square && panel0 + col[u] <= q0 + rb))
                        continue;
#pragma unroll
                    for (int w = 0; w < W; ++w)
                        atomicAdd(&racc[w * kPanelCols + col[u]], v[rb][u][w]);
                }
            }
        }
        // the batch's rows that are run records themselves: their row of the other table over the panel's columns
        for (uint32_t rb = 0; rb < nrows; ++rb) {
            const uint32_t q = q0 + rb;
            uint32_t *racc = bacc + rb * W * kPanelCols;
            const uint32_t hq = rp.index[q];
            if (hq != 0xFFFFFFFFu)
                for (uint32_t k0 = t; k0 < pcols; k0 += nt * UNR) {
                    uint32_t v[UNR][W];
#pragma unroll
                    for (uint32_t u = 0; u < UNR; ++u)
#pragma unroll
                        for (int w = 0; w < W; ++w)   // (a clamped column: see above)
                            v[u][w] = rp.corr[((size_t)w * rp.n_run + hq) * rp.n + panel0 + min(k0 + nt * u, pcols - 1u)];
#pragma unroll
                    for (uint32_t u = 0; u < UNR; ++u) {
                        const uint32_t k = k0 + nt * u;
                        if (k >= pcols || (square && panel0 + k <= q))
                            continue;
#pragma unroll
                        for (int w = 0; w < W; ++w)
                            atomicAdd(&racc[w * kPanelCols + k], v[u][w]);
                    }
                }
        }
    };
    // ---- hybrid path: the dense kernels' tallies of the hot columns (one packed word group per pair, canonical order) go
    // into the accumulators like events, on the event side of the batch barrier.  Inside the output loop they were a global
    // load between the result stores — gfx950 makes such a load wait for every store issued before it — and kept every
    // group on do_pair's slow form: the pair kernel of a clade-structured 50,000 x 30,000 took 4.8 ms for 15 GB.
    auto apply_hot = [&](uint32_t b, uint32_t t, uint32_t nt) {   // thread t of nt
        if (!hot)
            return;
        const uint32_t q0 = tile.i0 + b * RB, nrows = min((uint32_t)RB, tile.i1 - q0);
        uint32_t *bacc = acc + (UNI ? 0u : b & 1u) * ACC;
        constexpr uint32_t UH = W == 1 ? 8 : 4;
        for (uint32_t rb = 0; rb < nrows; ++rb) {
            const uint32_t q = q0 + rb;
            const uint64_t row_at = square ? (tri_row_start(n_cols, q) - out_base) - (uint64_t)(q + 1)
                                           : (uint64_t)(q - row_begin) * n_cols;
            uint32_t *racc = bacc + rb * W * kPanelCols;
            const uint32_t kfirst = square && q + 1u > panel0 ? min(q + 1u - panel0, pcols) : 0u;   // columns past the diagonal
            for (uint32_t k0 = kfirst + t; k0 < pcols; k0 += nt * UH) {
                uint32_t v[UH][W];
#pragma unroll
                for (uint32_t u = 0; u < UH; ++u) {   // (clamped column, masked below: no predicated loads)
#pragma unroll
                    for (int w = 0; w < W; ++w)
                        v[u][w] = 0;
                    P::add_hot(v[u], hot, row_at + panel0 + min(k0 + nt * u, pcols - 1u));
                }
#pragma unroll
                for (uint32_t u = 0; u < UH; ++u) {
                    const uint32_t k = k0 + nt * u;
                    if (k >= pcols)
                        continue;
#pragma unroll
                    for (int w = 0; w < W; ++w)
                        if (v[u][w])
                            atomicAdd(&racc[w * kPanelCols + k], v[u][w]);
                }
            }
        }
    };
    // ---- C of batch b from its accumulator buffer: constants, finalisation, canonical-order store
    auto output_batch = [&](const uint32_t b) {
        const uint32_t q0 = tile.i0 + b * RB;
        const uint32_t nrows = min((uint32_t)RB, tile.i1 - q0);
        for (uint32_t rb = 0; rb < nrows; ++rb) {
            const uint32_t q = q0 + rb;
            uint32_t aq[W];
#pragma unroll
            for (int w = 0; w < W; ++w)
                aq[w] = row_a[(size_t)w * row_npad + q] + fw.w[w];
            uint4 qc = make_uint4(0, 0, 0, 0);
            if constexpr (OUT == DST_TN93)
                qc = reinterpret_cast<const uint4 *>(q_counts)[q];
            const uint64_t row_at = square ? (tri_row_start(n_cols, q) - out_base) - (uint64_t)(q + 1)
                                           : (uint64_t)(q - row_begin) * n_cols;
            uint32_t *racc = acc + (UNI ? 0u : b & 1u) * ACC + rb * W * kPanelCols;
            // two adjacent results of row q: columns panel0 + ks and panel0 + ks + 1 (ks may be -1 .. pcols - 1:
            // ALIGNED pairs straddle the panel's edges); j: the thread's register copy of A(column) (HOIST)
            // cav / tcv: the hoisted A(column) words and packed base counts of the two columns (nullptr when ALIGNED)
            auto do_pair = [&](int32_t ks, const uint32_t (*cav2)[W], const uint32_t (*tcv)[TCW]) {
                bool in[2], live[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    in[h] = (uint32_t)(ks + h) < pcols;
                    live[h] = in[h] && (!square || panel0 + (uint32_t)(ks + h) > q);
                }
                if (!live[1] && !live[0])
                    return;  // outside the panel, or square: at or below the diagonal (accumulators stay 0 there)
                const uint64_t at = row_at + panel0 + (uint64_t)(int64_t)ks;   // modulo 2^64: right wherever live
                uint32_t o[2][NT];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t k = in[h] ? (uint32_t)(ks + h) : 0u;
                    uint32_t tot[W];
#pragma unroll
                    for (int w = 0; w < W; ++w) {
                        const uint32_t a = racc[w * kPanelCols + k];
                        if (a && in[h])
                            racc[w * kPanelCols + k] = 0;
                        uint32_t cav;
                        if constexpr (ALIGNED)
                            cav = cola[k];
                        else
                            cav = cav2[h][w];
                        tot[w] = a + cav + aq[w];
                    }
                    P::unpack(tot, o[h]);
                }
                if constexpr (OUT == OUT_INT) {
                    int64_t *out = static_cast<int64_t *>(out_v);
                    if (live[0] && live[1]) {
                        store_result2(out + at, (int64_t)o[0][0], (int64_t)o[1][0]);
                    } else if (live[0]) {
                        store_result(out + at, (int64_t)o[0][0]);
                    } else {
                        store_result(out + at + 1, (int64_t)o[1][0]);
                    }
                } else if constexpr (OUT == OUT_TALLY) {
                    uint32_t *out = static_cast<uint32_t *>(out_v);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if (live[h])
#pragma unroll
                            for (int x = 0; x < NT; ++x)
                                store_result(out + (at + h) * NT + x, o[h][x]);
                } else if constexpr (OUT == OUT_TALLY16) {
                    uint16_t *out = static_cast<uint16_t *>(out_v);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if (live[h])
#pragma unroll
                            for (int x = 0; x < NT; ++x)
                                store_result(out + (at + h) * NT + x, (uint16_t)o[h][x]);
                } else {
                    double *out = static_cast<double *>(out_v);
                    double d[2] = {0.0, 0.0};
                    auto fin = [&](int h) {
                        if (live[h]) {
                            uint4 tc = make_uint4(0, 0, 0, 0);
                            if constexpr (HOIST_TC) {
                                if constexpr (WIDE)
                                    tc = make_uint4(tcv[h][0], tcv[h][1], tcv[h][2], tcv[h][3]);
                                else
                                    tc = make_uint4(tcv[h][0] & 0xFFFFu, tcv[h][0] >> 16, tcv[h][1] & 0xFFFFu, tcv[h][1] >> 16);
                            } else if constexpr (OUT == DST_TN93) {
                                tc = reinterpret_cast<const uint4 *>(t_counts)[panel0 + (uint32_t)(ks + h)];
                            }
                            d[h] = finalize_pair<OUT>(o[h], qc, tc, LOGS ? logtab : kLogTab);
                        }
                    };
                    fin(0);   // (two copies of the formula: the two results' dependency chains interleave)
                    fin(1);
                    if (live[0] && live[1]) {
                        store_result2(out + at, d[0], d[1]);
                    } else if (live[0]) {
                        store_result(out + at, d[0]);
                    } else {
                        store_result(out + at + 1, d[1]);
                    }
                }
            };
            // Two adjacent results whose columns k, k + 1 are known to lie inside the panel and past the diagonal (a
            // whole group of a wave does): none of do_pair's per-lane tests, which cost ~50 scalar instructions per
            // pair of results (exec-mask bookkeeping) — more than the one scalar unit of a CU keeps up with at the
            // write rate.  c0 / c1: A(column) words; t0 / t1: the columns' base counts (tn93).
            auto fast_pair = [&](uint32_t k, const uint32_t *c0, const uint32_t *c1, uint4 tc0, uint4 tc1) {
                uint32_t w0[W], w1[W];
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    w0[w] = racc[w * kPanelCols + k] + c0[w] + aq[w];
                    w1[w] = racc[w * kPanelCols + k + 1] + c1[w] + aq[w];
                    racc[w * kPanelCols + k] = 0;          // (unconditional: cheaper than testing)
                    racc[w * kPanelCols + k + 1] = 0;
                }
                uint32_t o0[NT], o1[NT];
                P::unpack(w0, o0);
                P::unpack(w1, o1);
                const uint64_t at = row_at + panel0 + k;
                if constexpr (OUT == OUT_INT) {
                    store_result2(static_cast<int64_t *>(out_v) + at, (int64_t)o0[0], (int64_t)o1[0]);
                } else if constexpr (OUT == OUT_TALLY) {
                    uint32_t *out = static_cast<uint32_t *>(out_v) + at * NT;
#pragma unroll
                    for (int x = 0; x < NT; ++x) {
                        store_result(out + x, o0[x]);
                        store_result(out + NT + x, o1[x]);
                    }
                } else if constexpr (OUT == OUT_TALLY16) {
                    uint16_t *out = static_cast<uint16_t *>(out_v) + at * NT;
#pragma unroll
                    for (int x = 0; x < NT; ++x) {
                        store_result(out + x, (uint16_t)o0[x]);
                        store_result(out + NT + x, (uint16_t)o1[x]);
                    }
                } else {
                    const double d0 = finalize_pair<OUT>(o0, qc, tc0, LOGS ? logtab : kLogTab);
                    const double d1 = finalize_pair<OUT>(o1, qc, tc1, LOGS ? logtab : kLogTab);
                    store_result2(static_cast<double *>(out_v) + at, d0, d1);
                }
            };
            // is the wave's group of 128 columns from k0 on (uniform) wholly inside the panel and past the diagonal?
            auto whole_group = [&](int32_t k0) {
                return k0 >= 0 && (uint32_t)k0 + 128u <= pcols && (!square || panel0 + (uint32_t)k0 > q);
            };
            if constexpr (ALIGNED) {
                // The row's 2,048 results of this panel start `sh` elements into a 128-byte line.  The 16 (sh == 0)
                // or 17 lines-of-1-KB groups from that line's start go to the output waves as contiguous runs:
                // a wave's stores are whole, consecutive lines; only the panel's two edge lines are shared with
                // the neighbouring tiles.
                constexpr uint32_t ELEM = OUT == OUT_TALLY ? 4u * NT : OUT == OUT_TALLY16 ? 2u * NT : 8u;
                const uint32_t sh = (uint32_t)((reinterpret_cast<uintptr_t>(out_v) / ELEM + row_at + panel0) & 15u);
                // 17 groups over NOW waves: BASE each, the first REM waves (in an order rotating with the row) one more
                constexpr uint32_t GROUPS = kPanelCols / 128 + 1, BASE = GROUPS / NOW, REM = GROUPS % NOW;
                const uint32_t wv = tid >> 6, idx = (wv + NOW - q % NOW) % NOW;
                const uint32_t g0 = idx * BASE + min(idx, REM);
                auto do_group = [&](uint32_t g) {
                    const int32_t k0 = (int32_t)(128u * g) - (int32_t)sh;          // the group's first column (uniform)
                    if (!whole_group(k0)) {
                        do_pair(k0 + (int32_t)(2u * lane), nullptr, nullptr);
                        return;
                    }
                    const uint32_t k = (uint32_t)k0 + 2u * lane;
                    const uint4 none = make_uint4(0, 0, 0, 0);
                    fast_pair(k, &cola[k], &cola[k + 1], none, none);
                };
#pragma unroll
                for (uint32_t g = 0; g < BASE; ++g)
                    do_group(g0 + g);
                if (idx < REM)
                    do_group(g0 + BASE);
            }
            // panel-relative mapping: slot j of this wave = columns 2 (tid's wave x 64) + 2 OT j .. + 128
            auto do_slot = [&](int j, const uint32_t (*cav2)[W], const uint32_t (*tcv)[TCW]) {
                const int32_t k0 = (int32_t)(2u * (tid & ~63u) + 2u * OT * (uint32_t)j);
                if (!whole_group(k0)) {
                    do_pair(k0 + (int32_t)(2u * lane), cav2, tcv);
                    return;
                }
                uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0;
                if constexpr (HOIST_TC) {
                    if constexpr (WIDE) {
                        t0 = make_uint4(tcv[0][0], tcv[0][1], tcv[0][2], tcv[0][3]);
                        t1 = make_uint4(tcv[1][0], tcv[1][1], tcv[1][2], tcv[1][3]);
                    } else {
                        t0 = make_uint4(tcv[0][0] & 0xFFFFu, tcv[0][0] >> 16, tcv[0][1] & 0xFFFFu, tcv[0][1] >> 16);
                        t1 = make_uint4(tcv[1][0] & 0xFFFFu, tcv[1][0] >> 16, tcv[1][1] & 0xFFFFu, tcv[1][1] >> 16);
                    }
                }
                fast_pair((uint32_t)k0 + 2u * lane, cav2[0], cav2[1], t0, t1);
            };
            if constexpr (ALIGNED) {
            } else if constexpr (OUT == DST_TN93) {
                // one copy of the pair's code (two of the formula), not PAIRS: tn93's registers are at the limit the
                // LDS leaves (128); the hoisted values of slot j are picked by selects, not by indexing
#pragma unroll 1
                for (int j = 0; j < PAIRS; ++j) {
                    uint32_t cc[2][W], tt[2][TCW];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int w = 0; w < W; ++w) {
                            cc[h][w] = ca[0][h][w];
#pragma unroll
                            for (int jj = 1; jj < PAIRS; ++jj)
                                cc[h][w] = j == jj ? ca[jj][h][w] : cc[h][w];
                        }
#pragma unroll
                        for (int x = 0; x < TCW; ++x) {
                            tt[h][x] = tcp[0][h][x];
#pragma unroll
                            for (int jj = 1; jj < PAIRS; ++jj)
                                tt[h][x] = j == jj ? tcp[jj][h][x] : tt[h][x];
                        }
                    }
                    do_slot(j, cc, tt);
                }
            } else {
#pragma unroll
                for (int j = 0; j < PAIRS; ++j)
                    do_slot(j, ca[j], tcp[0]);
            }
        }
    };
    // Workgroup barrier for the LDS accumulators only.  __syncthreads() is also a fence for global memory: hipcc puts
    // s_waitcnt vmcnt(0) in front of it, which makes the output waves wait at every batch until all their result
    // stores have landed in HBM and the event waves wait for the loads they have just issued for the coming batches.
    // Nothing in global memory is handed between the waves of this kernel, so only the LDS traffic (lgkmcnt) has to be
    // complete.  The two roles run SEPARATE loops with the same number of barriers (s_barrier counts arrivals, it
    // does not care where a wave stands): the event pipeline's registers are then not live in the output code and
    // the other way round — the kernel needs the larger of the two register sets, not their sum.
#define DST_BATCH_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    if constexpr (UNI) {
        for (uint32_t b = 0; b < nbatch; ++b) {
            // the batch's entries kEventLanes at a time, the next slice's table entries and the list entries of the one
            // after it loaded before this slice's events are applied
            const uint32_t run = rofs[min(b * RB + RB, trows)] - rofs[b * RB];
            if (run) {
                Inl s_cur = load_inl(load_entry(b, 0));
                Entry s_en = load_entry(b, kEventLanes);
                for (uint32_t first = 0; first < run; first += kEventLanes) {
                    const Inl s_nx = load_inl(s_en);
                    s_en = load_entry(b, first + 2 * kEventLanes);
                    apply_bucket(s_cur, b);
                    s_cur = s_nx;
                }
            }
            apply_runs(b, threadIdx.x, 64u * kBlockWaves);
            apply_hot(b, threadIdx.x, 64u * kBlockWaves);
            DST_BATCH_BARRIER();
            output_batch(b);
            DST_BATCH_BARRIER();
        }
    } else if (event_role) {
        const Inl inl_none{make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), 0u};
        Entry en_n2{0u, 0u, false};
        Inl in_cur = inl_none, in_nx = inl_none;
        constexpr bool do_events = true;
        if (do_events) {  // prologue: batch 0's and batch 1's table entries, batch 2's list entry
            in_cur = load_inl(load_entry(0, 0));
            in_nx = load_inl(load_entry(1, 0));
            en_n2 = load_entry(2, 0);
        }
        for (uint32_t step = 0; step <= nbatch; ++step) {
            if (step < nbatch && do_events) {
                // ---- B of batch `step` into buffer step & 1
                apply_bucket(in_cur, step);
                // entries beyond the ones the pipeline carries (long lists: diverse data, records with runs of N): slice by
                // slice, the next slice's table entries and the list entries of the one after it loaded before this slice's
                // events are applied (one after the other every slice was two memory latencies with nothing else going on)
                const uint32_t run = rofs[min(step * RB + RB, trows)] - rofs[step * RB];
                if (run > kEventLanes) {
#ifdef DST_DBG_PLAIN_SLICES
                    for (uint32_t first = kEventLanes; first < run; first += kEventLanes)
                        apply_bucket(load_inl(load_entry(step, first)), step);
#else
                    Inl s_cur = load_inl(load_entry(step, kEventLanes));
                    Entry s_en = load_entry(step, 2 * kEventLanes);
                    for (uint32_t first = kEventLanes; first < run; first += kEventLanes) {
                        const Inl s_nx = load_inl(s_en);
                        s_en = load_entry(step, first + 2 * kEventLanes);
                        apply_bucket(s_cur, step);
                        s_cur = s_nx;
                    }
#endif
                }
                apply_runs(step, tid, kEventLanes);
                apply_hot(step, tid, kEventLanes);
                // Rotate the pipeline FIRST — these copies read what the previous step's loads delivered, which has
                // had a whole step to arrive — and only then issue the next loads.  Left to itself hipcc issues the
                // loads first and copies at the end of the iteration, which needs s_waitcnt vmcnt(0) right behind the
                // loads it has just issued: no overlap at all.  pin() fixes the copies in place, the scheduling
                // barrier keeps the loads below them.
                in_cur = in_nx;
                Entry en_use = en_n2;
                pin(in_cur.lo.x), pin(in_cur.lo.y), pin(in_cur.lo.z), pin(in_cur.lo.w);
                pin(in_cur.hi.x), pin(in_cur.hi.y), pin(in_cur.hi.z), pin(in_cur.hi.w);
                pin(in_cur.meta), pin(en_use.e), pin(en_use.rb);
                __builtin_amdgcn_sched_barrier(0);
                in_nx = load_inl(en_use);
                en_n2 = load_entry(step + 3, 0);
            }
            DST_BATCH_BARRIER();
        }
    } else {
        for (uint32_t step = 0; step <= nbatch; ++step) {
            if (step >= 1)
                output_batch(step - 1);
            DST_BATCH_BARRIER();
        }
    }
#undef DST_BATCH_BARRIER
}

// =============================================================================================
// shared preparation (dst_upload_shared): every rank packs and lists 1/world of the records, ONE all-gather of
// fixed-size blocks carries the lists (and counts) to everybody, and each rank splices them into the set's CSR
// =============================================================================================
// the block's header, list lengths (and zero padding up to the layout's records per rank) from the rank's own pass
__global__ __launch_bounds__(256) void shared_block_kernel(uint32_t *__restrict__ block, SharedLayout lay, uint32_t rec_first,
                                                           uint32_t count, const uint32_t *__restrict__ cnt_cold,
                                                           const uint32_t *__restrict__ cnt_hot,
                                                           const uint32_t *__restrict__ off_local,
                                                           const unsigned long long *__restrict__ first_bad)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < lay.rmax)
        block[lay.cnt_at + i] = i < count ? cnt_cold[rec_first + i] + cnt_hot[rec_first + i] : 0u;
    if (i < 16) {
        const uint32_t total = off_local[count];
        const unsigned long long bad = *first_bad;
        const uint32_t hdr[4] = {total, total > lay.ent_cap ? 1u : 0u, (uint32_t)bad, (uint32_t)(bad >> 32)};
        block[i] = i < 4 ? hdr[i] : 0u;
    }
}

// every record's list length (and counts) from the gathered blocks, in record order; len_all[n] = 0 for the scan
__global__ __launch_bounds__(256) void shared_lengths_kernel(const uint32_t *__restrict__ gathered, SharedLayout lay, uint32_t n,
                                                             uint32_t *__restrict__ len_all, uint32_t *__restrict__ counts)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r > n)
        return;
    if (r == n) {
        len_all[n] = 0;
        return;
    }
    const uint32_t k = r / lay.rmax, i = r - k * lay.rmax;
    const uint32_t *blk = gathered + (size_t)k * lay.words;
    len_all[r] = blk[lay.cnt_at + i];
    if (counts)
        reinterpret_cast<uint4 *>(counts)[r] = reinterpret_cast<const uint4 *>(blk + lay.counts_at)[i];
}

// rank blockIdx.y's entries to their place in the CSR (its first record's offset)
// (ent_room: what rec_ent holds — when a rank's lists did not fit its block the offsets run past it, and the upload fails)
__global__ __launch_bounds__(256) void shared_entries_kernel(const uint32_t *__restrict__ gathered, SharedLayout lay, uint32_t n,
                                                             const uint32_t *__restrict__ rec_off, uint32_t *__restrict__ rec_ent,
                                                             uint32_t ent_room)
{
    const uint32_t k = blockIdx.y;
    const uint32_t *blk = gathered + (size_t)k * lay.words;
    const uint32_t total = min(blk[0], lay.ent_cap);
    const uint32_t base = rec_off[min((uint32_t)((uint64_t)k * lay.rmax), n)];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256)
        if (base + i < ent_room)
            rec_ent[base + i] = blk[lay.ent_at + i];
}

// where every record's list crosses each multiple of kBucketSites sites (what the fill passes note on their way): mark g
// of a record = the index of its first entry at or beyond site g * kBucketSites.  One wave = one record, a lane = an
// entry: it is the mark of every multiple that lies behind its predecessor's site and not behind its own.  (A binary
// search per (record, multiple) — six dependent loads for each of 1.5 M threads — took 42 us at 50,000 x 30,000.)
__global__ __launch_bounds__(256) void range_marks_kernel(const uint32_t *__restrict__ rec_off, const uint32_t *__restrict__ rec_ent,
                                                          uint32_t n, uint32_t npad, uint32_t n_ranges,
                                                          uint32_t *__restrict__ range_start, uint32_t ent_room)
{
    const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (r >= n)
        return;
    const uint32_t lo = min(rec_off[r], ent_room), hi = min(rec_off[r + 1], ent_room);
    uint32_t next = 0;   // marks [0, next) are written (the same in every lane)
    for (uint32_t base = lo; base < hi; base += 64u) {
        const uint32_t i = base + lane;
        const bool live = i < hi;
        // the multiples up to this entry's site: g <= site / kBucketSites; the predecessor covered those up to its own
        const uint32_t ent = rec_ent[min(i, hi - 1u)];   // (unconditional load)
        const uint32_t mine = live ? min((ent & kSiteMask) / kBucketSites + 1u, n_ranges) : 0u;
        uint32_t before = __shfl_up(mine, 1);
        if (lane == 0)
            before = next;
        before = max(before, next);   // (sites ascend: `mine` does too)
        if (live)
            for (uint32_t g = before; g < mine; ++g)
                range_start[(size_t)g * npad + r] = i;
        const uint32_t last = min(hi - base, 64u) - 1u;
        next = max(next, (uint32_t)__shfl(mine, last));
    }
    for (uint32_t g = next + lane; g < n_ranges; g += 64u)   // multiples behind the last entry
        range_start[(size_t)g * npad + r] = hi;
}

// what the host reads after a shared upload: [0] first invalid byte over all ranks, [1..8] the sample's statistics,
// [9] list entries of the whole set, [10] ranks whose lists did not fit their block, [11] the largest block's entries
__global__ __launch_bounds__(64) void shared_report_kernel(const uint32_t *__restrict__ gathered, SharedLayout lay,
                                                           const unsigned long long *__restrict__ stats, unsigned long long *report)
{
    if (threadIdx.x >= 1 && threadIdx.x <= 8)
        report[threadIdx.x] = stats[threadIdx.x - 1];
    if (threadIdx.x != 0)
        return;
    unsigned long long bad = ~0ull, total = 0, over = 0, biggest = 0;
    for (uint32_t k = 0; k < lay.world; ++k) {
        const uint32_t *blk = gathered + (size_t)k * lay.words;
        const unsigned long long b = (unsigned long long)blk[2] | (unsigned long long)blk[3] << 32;
        bad = b < bad ? b : bad;
        total += blk[0];
        over += blk[1];
        biggest = blk[0] > biggest ? blk[0] : biggest;
    }
    report[0] = bad;
    report[9] = total;
    report[10] = over;
    report[11] = biggest;
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
    const uint32_t samples = (uint32_t)std::min<size_t>(set.n, kRefSamples);
    hipLaunchKernelGGL(ref_sample_kernel<false>, dim3((unsigned)set.nchunks), dim3(512), 0, stream,
                       reinterpret_cast<const uint32_t *>(set.planes), nullptr, 0, (uint32_t)set.n, (uint32_t)set.len,
                       (uint32_t)set.nchunks, (uint32_t)set.npad, samples, set.ref.planes, set.ref.hot_planes,
                       set.ref.partials, nullptr, 0u, nullptr);
    return hipGetLastError();
}

hipError_t launch_ref_sample_bytes(const uint8_t *d_codes, size_t row_stride, const DeviceSet &set, hipStream_t stream,
                                   uint32_t *zero, size_t zero_words, unsigned long long *first_bad)
{
    const uint32_t samples = (uint32_t)std::min<size_t>(set.n, kRefSamples);
    hipLaunchKernelGGL(ref_sample_kernel<true>, dim3((unsigned)set.nchunks), dim3(1024), 0, stream, nullptr, d_codes, row_stride,
                       (uint32_t)set.n, (uint32_t)set.len, (uint32_t)set.nchunks, (uint32_t)set.npad, samples, set.ref.planes,
                       set.ref.hot_planes, set.ref.partials, zero, (uint32_t)zero_words, first_bad);
    return hipGetLastError();
}

hipError_t launch_hot_list(const DeviceSet &set, hipStream_t stream)
{
    hipLaunchKernelGGL(hot_list_kernel, dim3(1), dim3(1024), 0, stream, set.ref.hot_planes, (uint32_t)set.nchunks,
                       set.ref.hot_sites, set.ref.partials, (uint32_t)std::min<size_t>(set.n, kRefSamples),
                       reinterpret_cast<unsigned long long *>(set.ref.stats));
    return hipGetLastError();
}

hipError_t launch_compact(const DeviceSet &src, const uint32_t *hot_sites, uint32_t n_hot, DeviceSet &dst, hipStream_t stream)
{
    dim3 grid((unsigned)dst.nchunks, (unsigned)((dst.npad + 255) / 256));
    hipLaunchKernelGGL(compact_kernel, grid, dim3(256), 0, stream, reinterpret_cast<const uint32_t *>(src.planes),
                       (uint32_t)src.n, (uint32_t)src.nchunks, (uint32_t)src.npad, hot_sites, n_hot, (uint32_t)dst.nchunks,
                       (uint32_t)dst.npad, dst.planes);
    return hipGetLastError();
}

hipError_t launch_index(const DeviceSet &set, const uint4 *ref_planes, const uint4 *hot_planes, bool fill, bool skip_nclass,
                        uint32_t *rec, uint32_t *rec_ent, unsigned long long *total, hipStream_t stream, uint32_t *range_start)
{
    const unsigned blocks = (unsigned)((set.n + 31) / 32);
    if (fill)
        hipLaunchKernelGGL(index_kernel<true>, dim3(blocks), dim3(256), 0, stream, set.planes, ref_planes, hot_planes,
                           (uint32_t)set.n, (uint32_t)set.nchunks, (uint32_t)set.npad, skip_nclass ? 1 : 0, rec, rec_ent, total, range_start);
    else
        hipLaunchKernelGGL(index_kernel<false>, dim3(blocks), dim3(256), 0, stream, set.planes, ref_planes, hot_planes,
                           (uint32_t)set.n, (uint32_t)set.nchunks, (uint32_t)set.npad, skip_nclass ? 1 : 0, rec, rec_ent, total, range_start);
    return hipGetLastError();
}

hipError_t launch_slot_fill(const DeviceSet &set, const uint4 *ref_planes, const uint4 *hot_planes, bool without_hot,
                            uint32_t *rec_off, uint32_t *rec_ent, uint32_t *range_start, hipStream_t stream, size_t rec_begin,
                            size_t rec_end, uint32_t ent_cap)
{
    // waves of 8 records x 8 chunks when that makes enough of them, else 4 x 16 or 2 x 32 (see the kernel)
    const uint32_t first = (uint32_t)std::min(rec_begin, set.n), n = (uint32_t)std::min(rec_end, set.n);
    const uint32_t nch = (uint32_t)set.nchunks, npad = (uint32_t)set.npad, count = n - first;
    if (n <= first)
        return hipSuccess;
    auto go = [&](auto cln, auto wh) {
        constexpr uint32_t CLN = decltype(cln)::value;
        constexpr uint32_t per_block = 4u * (64u / CLN);
        hipLaunchKernelGGL((slot_fill_kernel<CLN, decltype(wh)::value>), dim3((count + per_block - 1) / per_block), dim3(256), 0, stream,
                           set.rec.pre_slots, set.planes, ref_planes, hot_planes, n, nch, npad, rec_off, rec_ent, range_start, first,
                           ent_cap, set.runs.active ? set.runs.index : nullptr, set.runs.active ? set.runs.state : nullptr);
    };
    using std::integral_constant;
    const uint32_t cln = count >= 32768 ? 8u : count >= 16384 ? 16u : 32u;
    if (without_hot) {
        if (cln == 8) go(integral_constant<uint32_t, 8>{}, std::true_type{});
        else if (cln == 16) go(integral_constant<uint32_t, 16>{}, std::true_type{});
        else go(integral_constant<uint32_t, 32>{}, std::true_type{});
    } else {
        if (cln == 8) go(integral_constant<uint32_t, 8>{}, std::false_type{});
        else if (cln == 16) go(integral_constant<uint32_t, 16>{}, std::false_type{});
        else go(integral_constant<uint32_t, 32>{}, std::false_type{});
    }
    return hipGetLastError();
}

hipError_t launch_site_buckets(const DeviceSet &set, uint32_t n_panels, uint32_t *ovf_total, hipStream_t stream)
{
    const uint32_t n_sites = (uint32_t)(set.nchunks * kChunkSites);
    hipLaunchKernelGGL(site_bucket_kernel, dim3((n_sites + kBucketSites - 1) / kBucketSites, n_panels), dim3(kBucketThreads), 0,
                       stream, set.rec.off, set.rec.ent, set.rec.range_start, (uint32_t)set.n, (uint32_t)set.npad, n_sites, set.site.inl, set.site.ent,
                       ovf_total);
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

hipError_t launch_exclusive_scan(uint32_t *data, size_t n, uint32_t *tmp, hipStream_t stream, const uint32_t *src0,
                                 const uint32_t *src1, uint32_t *zero, uint32_t n_zero)
{
    if (n == 0 && !n_zero)
        return hipSuccess;
    if (n <= kScanSmallMax) {
        hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, stream, data, (uint32_t)n, src0, src1, zero, zero ? n_zero : 0u);
        return hipGetLastError();
    }
    if (n <= kScanMidMax && src0 && data != src0 && data != src1 && (!zero || n_zero <= 1024)) {
        hipLaunchKernelGGL(scan_mid_kernel, dim3((unsigned)((n + 4095) / 4096)), dim3(1024), 0, stream, data, (uint32_t)n, src0, src1,
                           zero, zero ? n_zero : 0u);
        return hipGetLastError();
    }
    if (zero && n_zero) {
        const hipError_t ez = hipMemsetAsync(zero, 0, n_zero * sizeof(uint32_t), stream);
        if (ez != hipSuccess)
            return ez;
    }
    const size_t nb = (n + kScanPerBlock - 1) / kScanPerBlock;
    hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nb), dim3(256), 0, stream, data, n, tmp, src0, src1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || nb == 1)
        return e;
    e = launch_exclusive_scan(tmp, nb, tmp + nb + 1, stream);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(256), 0, stream, data, n, tmp);
    return hipGetLastError();
}

hipError_t launch_sum2_u32(const uint32_t *a0, const uint32_t *a1, size_t n, unsigned long long *totals, hipStream_t stream)
{
    hipLaunchKernelGGL(sum2_u32_kernel, dim3((unsigned)std::min<size_t>(256, (n + 1023) / 1024 + 1)), dim3(256), 0, stream, a0, a1, n, totals);
    return hipGetLastError();
}

hipError_t launch_report(const unsigned long long *first_bad, const unsigned long long *stats, uint32_t *cnt_cold,
                         uint32_t *cnt_hot, size_t n, unsigned long long *report, hipStream_t stream, const RunIndex *runs,
                         uint32_t max_run, unsigned long long *totals)
{
    RunsArg ra{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (runs && runs->cnt_run)
        ra = RunsArg{runs->cnt_run, runs->run_cold, runs->run_hot, runs->index, runs->ids, runs->state, max_run};
    if (cnt_cold && totals) {   // (totals: four words the sample kernel cleared with the counters)
        const unsigned blocks = (unsigned)std::min<size_t>(128, (n + 255) / 256);
        hipLaunchKernelGGL(list_totals_kernel, dim3(blocks), dim3(256), 0, stream, cnt_cold, cnt_hot, (uint32_t)n, ra, totals);
    }
    hipLaunchKernelGGL(report_kernel, dim3(1), dim3(1024), 0, stream, first_bad, stats, cnt_cold && totals ? totals : nullptr, cnt_cold,
                       cnt_hot, (uint32_t)n, report, ra);
    return hipGetLastError();
}

hipError_t launch_aconst(const DeviceSet &set, int family, bool wide, const ConsensusLut *d_lut, hipStream_t stream)
{
    RunConst rc{nullptr, 0, nullptr, nullptr, nullptr, 0};
    if (set.runs.active)
        rc = RunConst{set.runs.aent, set.runs.aent_cap / (kMaxWords * sizeof(uint32_t)), set.runs.index, set.runs.mask, set.runs.known,
                      (uint32_t)set.runs.mask_words};
    hipLaunchKernelGGL(aconst_kernel, dim3((unsigned)((set.n + 3) / 4)), dim3(256), 0, stream, set.rec.off,
                       set.rec.ent, d_lut, family, wide ? 1 : 0, family_words(family, wide),
                       (uint32_t)set.n, (uint32_t)set.npad, set.aconst, rc);
    return hipGetLastError();
}

// the run records' tables of one (family, packing) from the set's current lists: the panels' first run records, the
// per-chunk sums in 7-bit pieces, then X's terms by the matrix cores (corr_mfma_kernel) and the run x run F terms;
// aconst_kernel must have run for the same (family, packing) first (it leaves the entries' a-words in runs.aent)
hipError_t launch_run_tables(const DeviceSet &set, int family, bool wide, bool without_hot, const ConsensusLut *d_lut, hipStream_t stream)
{
    const RunIndex &ru = set.runs;
    const uint32_t n = (uint32_t)set.n, n_panels = (uint32_t)((set.n + kPanelCols - 1) / kPanelCols);
    const uint32_t mw = (uint32_t)ru.mask_words, kpad = 32u * mw;
    const int words = family_words(family, wide);
    hipLaunchKernelGGL(run_panels_kernel, dim3((n_panels + 256) / 256), dim3(256), 0, stream, ru.ids, ru.n_run, n_panels, ru.panel_first);
    const size_t stride = ru.aent_cap / (kMaxWords * sizeof(uint32_t));
    const dim3 grid((n + 127) / 128, (ru.n_run + 127) / 128);   // corr_mfma_kernel: 128 records x 128 run records per block
#define DST_CORR(WW)                                                                                                                  \
    do {                                                                                                                              \
        const size_t lds = (size_t)4 * WW * kpad * sizeof(uint32_t);                                                                  \
        if (lds > 64 * 1024) {                                                                                                        \
            const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(chunk_sums_kernel<WW>),                          \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                          \
            if (ea != hipSuccess)                                                                                                     \
                return ea;                                                                                                            \
        }                                                                                                                             \
        /* the run COLUMNS' table from the plain sums, the run ROWS' table from the sums with the F terms folded in */              \
        hipLaunchKernelGGL((chunk_sums_kernel<WW>), dim3((n + 3) / 4), dim3(256), lds, stream, set.rec.off, set.rec.ent, ru.aent,     \
                           stride, n, kpad, ru.s7, nullptr, nullptr, nullptr, d_lut, family, wide ? 1 : 0);                           \
        hipLaunchKernelGGL((corr_mfma_kernel<WW, true>), grid, dim3(256), 0, stream, ru.mask + (size_t)ru.n_run * mw, mw, ru.s7, n, ru.n_run, ru.corr_t);     \
        hipLaunchKernelGGL((chunk_sums_kernel<WW>), dim3((n + 3) / 4), dim3(256), lds, stream, set.rec.off, set.rec.ent, ru.aent,     \
                           stride, n, kpad, ru.s7, ru.index, ru.mask, ru.known, d_lut, family, wide ? 1 : 0);                         \
        hipLaunchKernelGGL((corr_mfma_kernel<WW, false>), grid, dim3(256), 0, stream, ru.mask + (size_t)ru.n_run * mw, mw, ru.s7, n, ru.n_run, ru.corr);      \
    } while (0)
    switch (words) {
    case 1: DST_CORR(1); break;
    case 2: DST_CORR(2); break;
    case 3: DST_CORR(3); break;
    default: DST_CORR(4); break;
    }
#undef DST_CORR
    (void)without_hot;
    return hipGetLastError();
}

hipError_t launch_run_masks(const DeviceSet &set, hipStream_t stream)
{
    const uint32_t total = set.runs.n_run * (uint32_t)set.runs.mask_words;
    hipLaunchKernelGGL(run_masks_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, set.rec.pre_slots, set.runs.ids, set.runs.n_run,
                       (uint32_t)set.nchunks, (uint32_t)set.npad, (uint32_t)set.runs.mask_words, set.runs.mask,
                       set.runs.mask + (size_t)set.runs.n_run * set.runs.mask_words);
    return hipGetLastError();
}

hipError_t launch_run_known(const DeviceSet &set, const uint4 *ref_planes, const uint4 *hot_planes, hipStream_t stream)
{
    hipLaunchKernelGGL(run_known_kernel, dim3((unsigned)((set.nchunks + 255) / 256)), dim3(256), 0, stream, ref_planes, hot_planes,
                       (uint32_t)set.nchunks, set.runs.known);
    return hipGetLastError();
}

namespace {

template <int FAM, bool WIDE, int OUT, int EW>
hipError_t launch_cpair_ew(const ConsensusLaunch &cl, const FWords &fw, hipStream_t stream)
{
    const size_t smem = cpair_smem_words<FAM, WIDE, OUT, EW>() * sizeof(uint32_t) +
                        (OUT == DST_JC69 || OUT == DST_K80 || OUT == DST_TN93 ? 128 * sizeof(LogEntry) : 0);
    RunPair rp{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    if (cl.square && cl.cols->runs.active && cl.cols->runs.n_run)
        rp = RunPair{cl.cols->runs.index, cl.cols->runs.ids, cl.cols->runs.panel_first, cl.cols->runs.corr, cl.cols->runs.corr_t,
                     cl.cols->runs.n_run, (uint32_t)cl.cols->n};
    auto kern = consensus_pair_kernel<FAM, WIDE, OUT, EW>;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess)
            return e;
    }
    hipLaunchKernelGGL(kern, dim3(cl.ntiles), dim3(64 * kBlockWaves), smem, stream, cl.rows->rec.off, cl.rows->rec.ent,
                       cl.rows->aconst, (uint32_t)cl.rows->npad, cl.cols->site.inl, cl.cols->site.ent,
                       cl.cols->aconst, (uint32_t)cl.cols->npad, (uint32_t)(cl.cols->nchunks * kChunkSites), cl.d_lut,
                       fw, cl.d_tiles, cl.d_out, cl.rows->counts, cl.cols->counts, (uint32_t)cl.cols->n,
                       (uint32_t)cl.row_begin, cl.out_base, cl.square ? 1 : 0, cl.d_hot, rp);
    return hipGetLastError();
}

template <int FAM, bool WIDE, int OUT>
hipError_t launch_cpair(const ConsensusLaunch &cl, const FWords &fw, hipStream_t stream)
{
    if (cl.heavy_events >= 2)
        return launch_cpair_ew<FAM, WIDE, OUT, kEventWavesAll>(cl, fw, stream);
    if (cl.heavy_events)
        return launch_cpair_ew<FAM, WIDE, OUT, kEventWavesHeavy>(cl, fw, stream);
    return launch_cpair_ew<FAM, WIDE, OUT, event_waves<FAM, WIDE, OUT>()>(cl, fw, stream);
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

hipError_t launch_shared_block(uint32_t *block, const SharedLayout &lay, const DeviceSet &set, size_t rec_begin, size_t count,
                               const uint32_t *off_local, const unsigned long long *first_bad, hipStream_t stream)
{
    hipLaunchKernelGGL(shared_block_kernel, dim3((lay.rmax + 255) / 256), dim3(256), 0, stream, block, lay, (uint32_t)rec_begin,
                       (uint32_t)count, set.rec.pre_cold, set.rec.pre_hot, off_local, first_bad);
    return hipGetLastError();
}

hipError_t launch_shared_splice(const uint32_t *gathered, const SharedLayout &lay, DeviceSet &set, uint32_t *len_all,
                                uint32_t *scan_tmp, bool with_counts, unsigned long long *report, hipStream_t stream)
{
    const uint32_t n = (uint32_t)set.n, npad = (uint32_t)set.npad;
    const uint32_t ent_room = (uint32_t)std::min<size_t>(set.rec.ent_cap / sizeof(uint32_t), 0xFFFFFFFFu);
    hipLaunchKernelGGL(shared_lengths_kernel, dim3((n + 256) / 256), dim3(256), 0, stream, gathered, lay, n, len_all,
                       with_counts ? set.counts : nullptr);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = launch_exclusive_scan(set.rec.off, (size_t)n + 1, scan_tmp, stream, len_all);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(shared_entries_kernel, dim3(std::max(1u, std::min(256u, lay.ent_cap / 1024u + 1u)), lay.world), dim3(256), 0,
                       stream, gathered, lay, n, set.rec.off, set.rec.ent, ent_room);
    const uint32_t n_ranges = (uint32_t)((set.nchunks * kChunkSites + kBucketSites - 1) / kBucketSites);
    hipLaunchKernelGGL(range_marks_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, set.rec.off, set.rec.ent, n, npad, n_ranges,
                       set.rec.range_start, ent_room);
    hipLaunchKernelGGL(shared_report_kernel, dim3(1), dim3(64), 0, stream, gathered, lay,
                       reinterpret_cast<const unsigned long long *>(set.ref.stats), report);
    return hipGetLastError();
}

// every measurement macro this translation unit was compiled with (dst_build_flags): a production build has none
const char *consensus_build_flags()
{
    return ""
#ifdef DST_DBG_ALIGN_JC69
           " DST_DBG_ALIGN_JC69"
#endif
#ifdef DST_DBG_EVWAVES
           " DST_DBG_EVWAVES"
#endif
#ifdef DST_DBG_HEAVY_EW
           " DST_DBG_HEAVY_EW"
#endif
#ifdef DST_DBG_OLDMAP
           " DST_DBG_OLDMAP"
#endif
#ifdef DST_DBG_PLAIN_SLICES
           " DST_DBG_PLAIN_SLICES"
#endif
#ifdef DST_DBG_PLAIN_STORES
           " DST_DBG_PLAIN_STORES"
#endif
#ifdef DST_DBG_RB
           " DST_DBG_RB"
#endif
        ;
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
