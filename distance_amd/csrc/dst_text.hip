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
#include "dst_ctx.h"

#include <algorithm>
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

// `{:.12}` of v into out (at most 31 characters); false: not representable here (|v| >= 1.8e7)
__device__ __forceinline__ bool put_fixed12(double v, char *out, int &len)
{
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

__global__ __launch_bounds__(256) void number_kernel(const void *__restrict__ results, int is_int, SlabShape sh,
                                                     const uint32_t *__restrict__ row_id_off,
                                                     const uint32_t *__restrict__ col_id_off, int swap_ids,
                                                     NumText *__restrict__ nums, uint32_t *__restrict__ lens,
                                                     uint32_t *__restrict__ unsupported)
{
    uint32_t row, col;
    uint64_t p;
    if (!slab_pair(sh, row, col, p))
        return;
    NumText t;
    int len = 0;
    bool ok = true;
    if (is_int) {
        const long long v = static_cast<const long long *>(results)[p];
        if (v < 0) {
            t.c[0] = '-';
            len = 1 + put_u64((uint64_t)(-(v + 1)) + 1, t.c + 1);
        } else {
            len = put_u64((uint64_t)v, t.c);
        }
    } else {
        ok = put_fixed12(static_cast<const double *>(results)[p], t.c, len);
    }
    if (!ok) {
        atomicOr(unsupported, 1u);
        len = 0;
    }
    t.len = (uint8_t)len;
    reinterpret_cast<uint4 *>(nums)[2 * p] = reinterpret_cast<const uint4 *>(&t)[0];
    reinterpret_cast<uint4 *>(nums)[2 * p + 1] = reinterpret_cast<const uint4 *>(&t)[1];
    (void)swap_ids;
    lens[p] = (row_id_off[row + 1] - row_id_off[row]) + (col_id_off[col + 1] - col_id_off[col]) + (uint32_t)len + 3u;
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
    int rc = ensure_bytes(ctx, &ctx->text_res, &ctx->text_res_bytes, pairs * 8);
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&ctx->text_num, &ctx->text_num_bytes, pairs * 32);
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&ctx->text_len, &ctx->text_len_bytes, (pairs + 1) * sizeof(uint32_t));
    if (!rc)
        rc = ensure_bytes(ctx, (void **)&ctx->text_scan, &ctx->text_scan_bytes, scan_tmp_words(pairs + 1) * sizeof(uint32_t));
    if (!rc && !ctx->text_flag)
        HIP_TRY(ctx, hipMalloc((void **)&ctx->text_flag, 2 * sizeof(uint32_t)));
    if (!rc && !ctx->d_total)
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_total, 2 * sizeof(unsigned long long)));
    if (rc)
        return rc;
    rc = run_sets(ctx, measure, square, rows, cols, rb, re, DST_OUT_DISTANCE, ctx->text_res, pairs * 8, stream);
    if (rc)
        return rc;
    SlabShape sh{cols.n, rb, square ? square_row_start(cols.n, rb) : 0, square ? 1 : 0};
    const uint64_t widest = square ? cols.n - rb - 1 : cols.n;
    const dim3 grid((unsigned)((widest + 255) / 256), (unsigned)(re - rb));
    HIP_TRY(ctx, hipMemsetAsync(ctx->text_flag, 0, 2 * sizeof(uint32_t), stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->text_len + pairs, 0, sizeof(uint32_t), stream));
    hipLaunchKernelGGL(number_kernel, grid, dim3(256), 0, stream, ctx->text_res, measure_is_int(measure) ? 1 : 0, sh, rid.off,
                       cid.off, swap_ids, reinterpret_cast<NumText *>(ctx->text_num), ctx->text_len, ctx->text_flag);
    HIP_TRY(ctx, hipGetLastError());
    // the offsets are 32-bit: a slab's text must stay below 4 GB (checked against the un-scanned total first)
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_total, 0, 2 * sizeof(unsigned long long), stream));
    HIP_TRY(ctx, launch_sum2_u32(ctx->text_len, ctx->text_len, pairs, ctx->d_total, stream));
    unsigned long long total = 0;
    uint32_t flag = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&total, ctx->d_total, sizeof total, hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipMemcpyAsync(&flag, ctx->text_flag, sizeof flag, hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    if (flag)
        return fail(ctx, DST_ERR_STATE, "a value of this slab has no short {:.12} text (|v| >= 1.8e7): format it on the host");
    if (total >= (1ull << 32))
        return fail(ctx, DST_ERR_ARG, "text slab too large (4 GB of text per call)");
    if (total > cap)
        return fail(ctx, DST_ERR_CAPACITY, "text buffer too small for the requested rows");
    rc = ensure_bytes(ctx, (void **)&ctx->text_buf, &ctx->text_buf_bytes, (size_t)total + 16);
    if (rc)
        return rc;
    HIP_TRY(ctx, launch_exclusive_scan(ctx->text_len, pairs + 1, ctx->text_scan, stream));
    hipLaunchKernelGGL(line_kernel, grid, dim3(256), 0, stream, sh, rid.off, rid.chars, cid.off, cid.chars, swap_ids,
                       reinterpret_cast<const NumText *>(ctx->text_num), ctx->text_len, ctx->text_buf);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->text_buf, (size_t)total, hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    *len = (size_t)total;
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
