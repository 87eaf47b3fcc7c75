// dst_text.hip — gather_write()'s TSV text (src/lib.rs:612-644) produced on the device: "id1\tid2\tvalue\n" per pair in
// canonical order, `{}` for the integer measures and `{:.12}` for the others (Rust's fixed-precision Display prints
// the EXACT binary value rounded half-to-even at the 12th decimal, "NaN" / "inf" / "-inf", keeps the sign of -0.0).
//
// Why on the device: one MI355X computes the 1.25e9 distances of a 50,000-record alignment in 3 ms; turning them
// into 50 GB of text took the 16 host threads of the box 4 s (the CLI's formatter pool).  Here the host only writes.
//
//   number_kernel   one thread per pair: the value's text into a 32-byte record (31 characters + length) and the
//                   line's length (id1 + id2 + number + 3) — exact: 64x64->128-bit product m * 10^12, shift with
//                   round-half-even, then decimal digits
//   (scan)          exclusive scan of the line lengths -> offsets (launch_exclusive_scan)
//   line_kernel     one thread per pair: the line's bytes at its offset
//
// Values whose text the 32-byte record cannot hold (|v| >= 1.8e7 — no distance gets there) raise a flag and the call
// returns DST_ERR_STATE: the caller formats that slab on the host (dst_format_f64).
//
// Byte identity with the reference for jc69 / k80 / tn93 (src/measures.rs:76, 109-112, 187 call f64::ln = the host's
// libm; gather_write prints `{:.12}` of THAT value, src/lib.rs:626-633).  The device's logarithm is within an ulp of
// glibc's, not equal to it, so a device-finalised value can round the other way at the 12th decimal about once in 1e7
// lines.  For those measures the pair kernels therefore hand over the integer TALLIES (bit-exact) and number_kernel
// finalises them itself; a value within kGuardShift's relative distance (2^-47 |v|: 32-64 ulp, several times what the
// device and the host can differ by — tests/test_gpu_text_identity.py measures <= 4 ulp) of a rounding boundary of the
// 12th decimal is a NEAR TIE: its pair, tallies and place in the text go onto a short list (~1e-4 of the lines of a
// SARS-CoV-2-like alignment), and the host re-finalises exactly those with dst_finalize (libm, the reference's
// operation order) and overwrites their digits in the text it has just received.  Every other line is provably the
// same text for any value within the guard.  raw needs none of this (fin_raw is the IEEE division bit for bit), nor
// do the integers.
#include "dst_ctx.h"
#include "dst_device.hpp"

#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

namespace dst {
namespace {

struct NumText {
    char c[31];
    uint8_t len;
};
static_assert(sizeof(NumText) == 32, "one 32-byte record per pair");

__device__ __forceinline__ int put_u64(uint64_t v, char *out)
{
    char tmp[20];
    int n = 0;
    do {
        const uint64_t q = v / 10;
        tmp[n++] = (char)('0' + (uint32_t)(v - q * 10));
        v = q;
    } while (v);
    for (int k = 0; k < n; ++k)
        out[k] = tmp[n - 1 - k];
    return n;
}

// relative half-width of the near-tie guard around a rounding boundary of the 12th decimal: 2^-kGuardShift |v|
constexpr int kGuardShift = 47;

// `{:.12}` of v into out (at most 31 characters); false: not representable here (|v| >= 1.8e7).
// near: v lies within 2^-kGuardShift |v| of a point where the 12th decimal's rounding turns over
__device__ __forceinline__ bool put_fixed12(double v, char *out, int &len, bool &near)
{
    near = false;
    const uint64_t bits = (uint64_t)__double_as_longlong(v);
    const uint32_t bexp = (uint32_t)((bits >> 52) & 0x7FF);
    const uint64_t frac = bits & 0x000FFFFFFFFFFFFFull;
    int n = 0;
    if (bexp == 0x7FF) {
        if (frac) {
            out[0] = 'N', out[1] = 'a', out[2] = 'N';
            len = 3;
            return true;
        }
        if (bits >> 63)
            out[n++] = '-';
        out[n] = 'i', out[n + 1] = 'n', out[n + 2] = 'f';
        len = n + 3;
        return true;
    }
    if (bits >> 63)
        out[n++] = '-';
    // |v| = m * 2^-sh exactly (normal: implicit leading one; subnormal: exponent of the smallest normal)
    const uint64_t m = bexp ? (frac | 0x0010000000000000ull) : frac;
    const int sh = 1075 - (int)(bexp ? bexp : 1);   // 1 .. 1074 for |v| < 2^52
    if (m != 0 && sh <= 0)
        return false;
    // R = round-half-even(m * 10^12 / 2^sh); P = m * 10^12 < 2^93 as (hi, lo)
    uint64_t R = 0;
    if (m != 0 && sh < 128) {
        const uint64_t lo = m * 1000000000000ull, hi = __umul64hi(m, 1000000000000ull);
        uint64_t r_hi, rem_hi, rem_lo, half_hi, half_lo;
        if (sh < 64) {
            R = (lo >> sh) | (hi << (64 - sh));     // sh >= 1
            r_hi = hi >> sh;
            rem_hi = 0;
            rem_lo = lo & ((1ull << sh) - 1);
            half_hi = 0;
            half_lo = 1ull << (sh - 1);
        } else {
            const int s2 = sh - 64;                 // 0 .. 63
            R = s2 ? hi >> s2 : hi;
            r_hi = 0;
            rem_hi = s2 ? hi & ((1ull << s2) - 1) : 0;
            rem_lo = lo;
            half_hi = s2 ? 1ull << (s2 - 1) : 0;
            half_lo = s2 ? 0 : 1ull << 63;
        }
        if (r_hi)
            return false;                           // |v| >= 2^64 / 10^12
        const bool above = rem_hi > half_hi || (rem_hi == half_hi && rem_lo > half_lo);
        const bool tie = rem_hi == half_hi && rem_lo == half_lo;
        {   // |rem - half| <= P >> kGuardShift (+1: the shift truncates), all in units of 2^-sh x 10^-12
            const uint64_t big_hi = above ? rem_hi : half_hi, big_lo = above ? rem_lo : half_lo;
            const uint64_t sml_hi = above ? half_hi : rem_hi, sml_lo = above ? half_lo : rem_lo;
            const uint64_t d_lo = big_lo - sml_lo, d_hi = big_hi - sml_hi - (big_lo < sml_lo ? 1u : 0u);
            const uint64_t guard = ((lo >> kGuardShift) | (hi << (64 - kGuardShift))) + 1;   // hi < 2^29: no bits lost
            near = d_hi == 0 && d_lo <= guard;
        }
        if (above || (tie && (R & 1))) {
            R += 1;
            if (R == 0)
                return false;
        }
    }
    const uint64_t ip = R / 1000000000000ull;
    uint64_t fp = R - ip * 1000000000000ull;
    n += put_u64(ip, out + n);
    out[n++] = '.';
    for (int k = 11; k >= 0; --k) {
        const uint64_t q = fp / 10;
        out[n + k] = (char)('0' + (uint32_t)(fp - q * 10));
        fp = q;
    }
    len = n + 12;
    return true;
}

// pair p of the slab = (row, column): blockIdx.y = row - row_begin, x = position in the row
struct SlabShape {
    uint64_t n_cols, row_begin, out_base;   // out_base: canonical index of the slab's first pair
    int square;
};

__device__ __forceinline__ bool slab_pair(const SlabShape &sh, uint32_t &row, uint32_t &col, uint64_t &p)
{
    row = (uint32_t)sh.row_begin + blockIdx.y;
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (sh.square) {
        col = row + 1 + x;
        if (col >= sh.n_cols)
            return false;
        p = (uint64_t)row * (2 * sh.n_cols - row - 1) / 2 - sh.out_base + x;
    } else {
        col = x;
        if (col >= sh.n_cols)
            return false;
        p = (uint64_t)blockIdx.y * sh.n_cols + x;
    }
    return true;
}

// one near tie: what the host needs to re-finalise the pair and to find its digits in the text
struct NearTie {
    uint32_t row, col;
    uint32_t at;        // number_kernel: the pair's index in the slab; after place_kernel: offset of the number in the text
    uint32_t pre_len;   // characters of the line before the number (id1 + id2 + 2)
    uint32_t t[4];      // the pair's tallies
};
static_assert(sizeof(NearTie) == 32, "");

// SRC: -1 the slab's int64 results, -2 its f64 results (raw), else a measure id: its tallies (T = uint16_t / uint32_t),
// finalised here — the same device arithmetic as the pair kernels' epilogue — with near ties noted for the host
template <int SRC, class T>
__global__ __launch_bounds__(256) void number_kernel(const void *__restrict__ results, SlabShape sh,
                                                     const uint32_t *__restrict__ row_id_off,
                                                     const uint32_t *__restrict__ col_id_off,
                                                     const uint32_t *__restrict__ row_counts,
                                                     const uint32_t *__restrict__ col_counts,
                                                     NumText *__restrict__ nums, uint32_t *__restrict__ lens,
                                                     uint32_t *__restrict__ flags, NearTie *__restrict__ ties,
                                                     uint32_t ties_cap)
{
    uint32_t row, col;
    uint64_t p;
    if (!slab_pair(sh, row, col, p))
        return;
    NumText t;
    int len = 0;
    bool ok = true, near = false;
    uint32_t o[4] = {0, 0, 0, 0};
    if constexpr (SRC == -1) {
        const long long v = static_cast<const long long *>(results)[p];
        if (v < 0) {
            t.c[0] = '-';
            len = 1 + put_u64((uint64_t)(-(v + 1)) + 1, t.c + 1);
        } else {
            len = put_u64((uint64_t)v, t.c);
        }
    } else if constexpr (SRC == -2) {
        ok = put_fixed12(static_cast<const double *>(results)[p], t.c, len, near);
        near = false;   // raw: the device's quotient IS the host's
    } else {
        constexpr int NT = SRC == DST_K80 ? 3 : SRC == DST_TN93 ? 4 : 2;
        const T *tl = static_cast<const T *>(results) + p * NT;
#pragma unroll
        for (int k = 0; k < NT; ++k)
            o[k] = tl[k];
        uint4 qc = make_uint4(0, 0, 0, 0), tc = qc;
        if constexpr (SRC == DST_TN93) {
            qc = reinterpret_cast<const uint4 *>(row_counts)[row];
            tc = reinterpret_cast<const uint4 *>(col_counts)[col];
        }
        ok = put_fixed12(finalize_pair<SRC, true>(o, qc, tc), t.c, len, near);
    }
    if (!ok) {
        atomicOr(&flags[0], 1u);
        len = 0;
    }
    t.len = (uint8_t)len;
    reinterpret_cast<uint4 *>(nums)[2 * p] = reinterpret_cast<const uint4 *>(&t)[0];
    reinterpret_cast<uint4 *>(nums)[2 * p + 1] = reinterpret_cast<const uint4 *>(&t)[1];
    const uint32_t pre = (row_id_off[row + 1] - row_id_off[row]) + (col_id_off[col + 1] - col_id_off[col]) + 2u;
    lens[p] = pre + (uint32_t)len + 1u;
    if (near && ok) {
        const uint32_t k = atomicAdd(&flags[1], 1u);
        if (k < ties_cap) {
            NearTie e;
            e.row = row;
            e.col = col;
            e.at = (uint32_t)p;
            e.pre_len = pre;
            e.t[0] = o[0], e.t[1] = o[1], e.t[2] = o[2], e.t[3] = o[3];
            reinterpret_cast<uint4 *>(ties)[2 * k] = reinterpret_cast<const uint4 *>(&e)[0];
            reinterpret_cast<uint4 *>(ties)[2 * k + 1] = reinterpret_cast<const uint4 *>(&e)[1];
        }
    }
}

// after the scan: where each near tie's number starts in the slab's text
__global__ __launch_bounds__(256) void place_kernel(NearTie *__restrict__ ties, uint32_t n, const uint32_t *__restrict__ offs)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < n)
        ties[k].at = offs[ties[k].at] + ties[k].pre_len;
}

__global__ __launch_bounds__(256) void line_kernel(SlabShape sh, const uint32_t *__restrict__ row_id_off,
                                                   const char *__restrict__ row_ids,
                                                   const uint32_t *__restrict__ col_id_off,
                                                   const char *__restrict__ col_ids, int swap_ids,
                                                   const NumText *__restrict__ nums, const uint32_t *__restrict__ offs,
                                                   char *__restrict__ text)
{
    uint32_t row, col;
    uint64_t p;
    if (!slab_pair(sh, row, col, p))
        return;
    char *o = text + offs[p];
    // id1 is the row's record unless the caller swaps (stream mode prints the loaded record first)
    const char *a = row_ids + row_id_off[row], *b = col_ids + col_id_off[col];
    uint32_t la = row_id_off[row + 1] - row_id_off[row], lb = col_id_off[col + 1] - col_id_off[col];
    if (swap_ids) {
        const char *ta = a;
        a = b;
        b = ta;
        const uint32_t tl = la;
        la = lb;
        lb = tl;
    }
    for (uint32_t k = 0; k < la; ++k)
        o[k] = a[k];
    o += la;
    *o++ = '\t';
    for (uint32_t k = 0; k < lb; ++k)
        o[k] = b[k];
    o += lb;
    *o++ = '\t';
    const NumText t = nums[p];
    for (uint32_t k = 0; k < t.len; ++k)
        o[k] = t.c[k];
    o += t.len;
    *o = '\n';
}

// the near ties of a slab whose text has arrived in `out`: re-finalise each on the host (libm, the reference's operation
// order: what the reference prints) and put its digits in place.  A number of another length (a tie at 9.99...: one in
// ~1e13 lines) moves the rest of the text.
int patch_near_ties(dst_ctx *ctx, int measure, int row_slot, int col_slot, int swap_ids, const NearTie *ties, uint32_t n_ties,
                    char *out, size_t cap, size_t *len)
{
    const uint32_t *rc = measure == DST_TN93 ? ctx->text_counts[row_slot].data() : nullptr;
    const uint32_t *cc = measure == DST_TN93 ? ctx->text_counts[col_slot].data() : nullptr;
    struct Moved {
        uint32_t at, old_len, new_len;
        char text[40];
    };
    std::vector<std::vector<Moved>> moved_by(1);
    std::vector<uint64_t> patched_by(1, 0);
    auto work = [&](size_t part, uint32_t k0, uint32_t k1) {
        for (uint32_t k = k0; k < k1; ++k) {
            const NearTie &e = ties[k];
            double f = 0;
            int64_t iv = 0;
            const uint32_t *q = rc ? rc + 4 * (size_t)e.row : nullptr, *t = cc ? cc + 4 * (size_t)e.col : nullptr;
            // record_1 is the one whose id is printed first (src/lib.rs:325, 432-434)
            dst_finalize(measure, e.t, swap_ids ? t : q, swap_ids ? q : t, &f, &iv);
            char text[40];
            const int new_len = dst_format_distance(measure, f, 0, text, sizeof text);
            char *at = out + e.at;
            uint32_t old_len = 0;
            while (at[old_len] != '\n')
                ++old_len;
            if ((uint32_t)new_len == old_len) {
                if (std::memcmp(at, text, old_len) != 0) {
                    std::memcpy(at, text, old_len);
                    patched_by[part] += 1;
                }
            } else {
                Moved m{e.at, old_len, (uint32_t)new_len, {}};
                std::memcpy(m.text, text, (size_t)new_len);
                moved_by[part].push_back(m);
            }
        }
    };
    const size_t parts = n_ties >= 32768 ? std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency())) : 1;
    if (parts == 1) {
        work(0, 0, n_ties);
    } else {
        moved_by.resize(parts);
        patched_by.resize(parts, 0);
        std::vector<std::thread> th;
        for (size_t k = 0; k < parts; ++k)
            th.emplace_back(work, k, (uint32_t)((uint64_t)n_ties * k / parts), (uint32_t)((uint64_t)n_ties * (k + 1) / parts));
        for (auto &t : th)
            t.join();
    }
    std::vector<Moved> moved;
    for (size_t k = 0; k < parts; ++k) {
        moved.insert(moved.end(), moved_by[k].begin(), moved_by[k].end());
        ctx->text_patched += patched_by[k];
    }
    if (!moved.empty()) {
        std::sort(moved.begin(), moved.end(), [](const Moved &x, const Moved &y) { return x.at < y.at; });
        long long growth = 0;
        for (const Moved &m : moved)
            growth += (long long)m.new_len - (long long)m.old_len;
        if ((long long)*len + growth > (long long)cap)
            return fail(ctx, DST_ERR_CAPACITY, "text buffer too small for the requested rows");
        // back to front, so that every offset still refers to the text as the device wrote it
        for (size_t k = moved.size(); k-- > 0;) {
            const Moved &m = moved[k];
            char *at = out + m.at;
            std::memmove(at + m.new_len, at + m.old_len, *len - (m.at + m.old_len));
            std::memcpy(at, m.text, m.new_len);
            *len = (size_t)((long long)*len + (long long)m.new_len - (long long)m.old_len);
        }
        ctx->text_patched += moved.size();
    }
    return DST_OK;
}

int text_common(dst_ctx *ctx, int measure, bool square, int row_slot, int col_slot, uint64_t rb, uint64_t re, int swap_ids,
                char *out, size_t cap, size_t *len)
{
    if (!ctx)
        return DST_ERR_ARG;
    if (!len || (!out && cap))
        return fail(ctx, DST_ERR_ARG, "null pointer");
    *len = 0;
    if (measure < DST_N || measure > DST_TN93)
        return fail(ctx, DST_ERR_ARG, "unknown measure");
    if (row_slot < 0 || row_slot > 1 || col_slot < 0 || col_slot > 1)
        return fail(ctx, DST_ERR_ARG, "slot must be 0 or 1");
    DeviceSet &rows = ctx->set[row_slot], &cols = ctx->set[col_slot];
    if (!rows.loaded || !cols.loaded)
        return fail(ctx, DST_ERR_STATE, "set not uploaded");
    dst_ctx::Ids &rid = ctx->ids[row_slot], &cid = ctx->ids[col_slot];
    if (!rid.off || !cid.off || rid.n != rows.n || cid.n != cols.n)
        return fail(ctx, DST_ERR_STATE, "record ids of the set not given (dst_set_ids)");
    if (rb > re || re > rows.n)
        return fail(ctx, DST_ERR_ARG, "row range out of bounds");
    const uint64_t pairs = pairs_in_rows(square, cols.n, rb, re);
    if (pairs == 0)
        return DST_OK;
    if (pairs >= (1ull << 31) || re - rb > 65535)
        return fail(ctx, DST_ERR_ARG, "text slab too large (at most 2^31 pairs and 65,535 rows per call)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = ctx->stream;
    // jc69 / k80 / tn93: from the tallies, with the near ties of the 12th decimal left to the host (see the top of the file)
    const bool from_tallies = measure == DST_JC69 || measure == DST_K80 || measure == DST_TN93;
    const bool tally16 = from_tallies && rows.len <= 65535;
    const int res_kind = !from_tallies ? DST_OUT_DISTANCE : tally16 ? DST_OUT_TALLY16 : DST_OUT_TALLY;
    const size_t res_bytes = dst_out_bytes(measure, res_kind, pairs);
    const uint32_t ties_cap = from_tallies ? (uint32_t)std::max<uint64_t>(4096, pairs / 16) : 0;
    int rc = ensure_bytes(ctx, &ctx->text_res, &ctx->text_res_bytes, res_bytes);
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&ctx->text_num, &ctx->text_num_bytes, pairs * 32);
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&ctx->text_len, &ctx->text_len_bytes, (pairs + 1) * sizeof(uint32_t));
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&ctx->text_scan, &ctx->text_scan_bytes, scan_tmp_words(pairs + 1) * sizeof(uint32_t));
    if (!rc && ties_cap)
        rc = ensure_bytes(ctx, &ctx->text_ties, &ctx->text_ties_bytes, (size_t)ties_cap * sizeof(NearTie));
    if (!rc && !ctx->text_flag)
        HIP_TRY(ctx, hipMalloc((void **)&ctx->text_flag, 2 * sizeof(uint32_t)));
    if (!rc && !ctx->d_total)
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_total, 2 * sizeof(unsigned long long)));
    if (rc)
        return rc;
    if (measure == DST_TN93) {
        // the base counts: on the device for number_kernel, on the host for the near ties
        for (int slot : {row_slot, col_slot}) {
            DeviceSet &s = ctx->set[slot];
            rc = need_counts(ctx, s, stream);
            if (rc)
                return rc;
            if (ctx->text_counts_epoch[slot] != s.epoch || ctx->text_counts[slot].size() != s.n * 4) {
                ctx->text_counts[slot].resize(s.n * 4);
                HIP_TRY(ctx, hipMemcpyAsync(ctx->text_counts[slot].data(), s.counts, s.n * 16, hipMemcpyDeviceToHost, stream));
                HIP_TRY(ctx, hipStreamSynchronize(stream));
                ctx->text_counts_epoch[slot] = s.epoch;
            }
        }
    }
    rc = run_sets(ctx, measure, square, rows, cols, rb, re, res_kind, ctx->text_res, res_bytes, stream);
    if (rc)
        return rc;
    SlabShape sh{cols.n, rb, square ? square_row_start(cols.n, rb) : 0, square ? 1 : 0};
    const uint64_t widest = square ? cols.n - rb - 1 : cols.n;
    const dim3 grid((unsigned)((widest + 255) / 256), (unsigned)(re - rb));
    HIP_TRY(ctx, hipMemsetAsync(ctx->text_flag, 0, 2 * sizeof(uint32_t), stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->text_len + pairs, 0, sizeof(uint32_t), stream));
    NumText *nums = reinterpret_cast<NumText *>(ctx->text_num);
    NearTie *d_ties = static_cast<NearTie *>(ctx->text_ties);
#define DST_NUMBER(SRC, T)                                                                                                  \
    hipLaunchKernelGGL((number_kernel<SRC, T>), grid, dim3(256), 0, stream, ctx->text_res, sh, rid.off, cid.off, rows.counts, \
                       cols.counts, nums, ctx->text_len, ctx->text_flag, d_ties, ties_cap)
    if (measure_is_int(measure))
        DST_NUMBER(-1, uint32_t);
    else if (measure == DST_RAW)
        DST_NUMBER(-2, uint32_t);
    else if (measure == DST_JC69 && tally16)
        DST_NUMBER(DST_JC69, uint16_t);
    else if (measure == DST_JC69)
        DST_NUMBER(DST_JC69, uint32_t);
    else if (measure == DST_K80 && tally16)
        DST_NUMBER(DST_K80, uint16_t);
    else if (measure == DST_K80)
        DST_NUMBER(DST_K80, uint32_t);
    else if (tally16)
        DST_NUMBER(DST_TN93, uint16_t);
    else
        DST_NUMBER(DST_TN93, uint32_t);
#undef DST_NUMBER
    HIP_TRY(ctx, hipGetLastError());
    // the offsets are 32-bit: a slab's text must stay below 4 GB (checked against the un-scanned total first)
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_total, 0, 2 * sizeof(unsigned long long), stream));
    HIP_TRY(ctx, launch_sum2_u32(ctx->text_len, ctx->text_len, pairs, ctx->d_total, stream));
    unsigned long long total = 0;
    uint32_t flags[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(&total, ctx->d_total, sizeof total, hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipMemcpyAsync(flags, ctx->text_flag, sizeof flags, hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    if (flags[0])
        return fail(ctx, DST_ERR_STATE, "a value of this slab has no short {:.12} text (|v| >= 1.8e7): format it on the host");
    const uint32_t n_ties = flags[1];
    if (n_ties > ties_cap)   // (distances far above 1: the guard is relative)
        return fail(ctx, DST_ERR_STATE, "too many values of this slab lie near a rounding boundary of the 12th decimal: format it on the host");
    if (total >= (1ull << 32))
        return fail(ctx, DST_ERR_ARG, "text slab too large (4 GB of text per call)");
    if (total > cap)
        return fail(ctx, DST_ERR_CAPACITY, "text buffer too small for the requested rows");
    rc = ensure_bytes(ctx, (void **)&ctx->text_buf, &ctx->text_buf_bytes, (size_t)total + 16);
    if (rc)
        return rc;
    if (n_ties && ctx->text_ties_host_bytes < (size_t)n_ties * sizeof(NearTie)) {
        if (ctx->text_ties_host)
            HIP_TRY(ctx, hipHostFree(ctx->text_ties_host));
        ctx->text_ties_host = nullptr;
        ctx->text_ties_host_bytes = 0;
        const size_t want = std::max<size_t>((size_t)n_ties * 2, 8192) * sizeof(NearTie);
        HIP_TRY(ctx, hipHostMalloc(&ctx->text_ties_host, want, hipHostMallocDefault));
        ctx->text_ties_host_bytes = want;
    }
    HIP_TRY(ctx, launch_exclusive_scan(ctx->text_len, pairs + 1, ctx->text_scan, stream));
    hipLaunchKernelGGL(line_kernel, grid, dim3(256), 0, stream, sh, rid.off, rid.chars, cid.off, cid.chars, swap_ids,
                       reinterpret_cast<const NumText *>(ctx->text_num), ctx->text_len, ctx->text_buf);
    HIP_TRY(ctx, hipGetLastError());
    if (n_ties) {
        hipLaunchKernelGGL(place_kernel, dim3((n_ties + 255) / 256), dim3(256), 0, stream, d_ties, n_ties, ctx->text_len);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(ctx->text_ties_host, d_ties, (size_t)n_ties * sizeof(NearTie), hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->text_buf, (size_t)total, hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    *len = (size_t)total;
    ctx->text_near_ties += n_ties;
    if (n_ties)
        return patch_near_ties(ctx, measure, row_slot, col_slot, swap_ids, static_cast<const NearTie *>(ctx->text_ties_host), n_ties,
                               out, cap, len);
    return DST_OK;
}

}  // namespace
}  // namespace dst

using namespace dst;

extern "C" {

int dst_set_ids(dst_ctx *ctx, int slot, const char *chars, const uint64_t *offsets, uint64_t n)
{
    if (!ctx)
        return DST_ERR_ARG;
    if (slot < 0 || slot > 1 || !offsets || (!chars && n && offsets[n]))
        return fail(ctx, DST_ERR_ARG, "bad argument");
    if (offsets[n] >= (1ull << 32))
        return fail(ctx, DST_ERR_ARG, "record ids longer than 4 GB in total");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    dst_ctx::Ids &ids = ctx->ids[slot];
    std::vector<uint32_t> off32(n + 1);
    for (uint64_t k = 0; k <= n; ++k) {
        if (k && offsets[k] < offsets[k - 1])
            return fail(ctx, DST_ERR_ARG, "id offsets must not decrease");
        off32[k] = (uint32_t)offsets[k];
    }
    int rc = ensure_bytes(ctx, (void **)&ids.off, &ids.off_bytes, (n + 1) * sizeof(uint32_t));
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&ids.chars, &ids.chars_bytes, std::max<size_t>(offsets[n], 1));
    if (rc)
        return rc;
    HIP_TRY(ctx, hipMemcpy(ids.off, off32.data(), (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (offsets[n])
        HIP_TRY(ctx, hipMemcpy(ids.chars, chars, offsets[n], hipMemcpyHostToDevice));
    ids.n = n;
    return DST_OK;
}

int dst_text_stats(const dst_ctx *ctx, uint64_t *near_ties, uint64_t *rewritten)
{
    if (!ctx)
        return DST_ERR_ARG;
    if (near_ties)
        *near_ties = ctx->text_near_ties;
    if (rewritten)
        *rewritten = ctx->text_patched;
    return DST_OK;
}

int dst_text_square(dst_ctx *ctx, int measure, uint64_t row_begin, uint64_t row_end, char *out, size_t capacity, size_t *len)
{
    return text_common(ctx, measure, true, 0, 0, row_begin, row_end, 0, out, capacity, len);
}

int dst_text_rect(dst_ctx *ctx, int measure, int row_slot, int col_slot, uint64_t row_begin, uint64_t row_end, int swap_ids,
                  char *out, size_t capacity, size_t *len)
{
    return text_common(ctx, measure, false, row_slot, col_slot, row_begin, row_end, swap_ids, out, capacity, len);
}

}  // extern "C"
