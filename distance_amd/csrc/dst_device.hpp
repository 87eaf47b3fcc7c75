// dst_device.hpp — device-side helpers shared by the pair kernels (dst_kernels.hip: dense bit-plane
// path, dst_consensus.hip: consensus-delta path): canonical-order indexing and the finalisation math
// (tallies -> f64 in the reference's operation order, src/measures.rs:68, 76, 109-112, 118-190).
// Both files are built with -ffp-contract=off: rustc never fuses a*b+c.
#pragma once
#include "dst_internal.h"
#include "dst_logtab.h"

namespace dst {
namespace {

__device__ __forceinline__ uint64_t tri_row_start(uint64_t n, uint64_t i)
{
    return i * (2 * n - i - 1) / 2;
}

// ---- natural logarithm for the fused finalisation ------------------------------------------------
// Table-driven, ~25 f64 instructions against ~90 for the library's: x = 2^k z, z in [0.6875, 1.375) cut into
// 128 pieces (tools/gen_logtab.py): ln x = k ln2 + logc + log1p(z/c - 1), |z/c - 1| <= 2^-7, Taylor to r^9,
// the head and the tail of the sum kept apart.  Within 1 ulp of glibc's log (which the reference calls through
// f64::ln) over [2^-1000, inf); ln(1) = +0 exactly; zeros, negatives, NaN, infinities and values below 2^-1000
// take the library's path, so NaN / -inf patterns are the library's.
struct LogEntry {
    double invc, logc;
};
__device__ const LogEntry kLogTab[128] = {DST_LOG_TABLE};

__device__ __attribute__((noinline)) double dst_log_special(double x) { return log(x); }

// `tab`: kLogTab, or a copy of it in LDS — the consensus pair kernel's output waves must not LOAD from global memory
// between their result stores (gfx950 counts loads and stores in one in-order counter: the load would wait for every
// store issued before it to land)
__device__ __forceinline__ double dst_log(double x, const LogEntry *tab = kLogTab)
{
    if (!(x >= 0x1p-1000 && x < __builtin_huge_val()))
        return dst_log_special(x);  // one out-of-line copy of the library's code per kernel, rarely run
    const uint64_t tmp = (uint64_t)__double_as_longlong(x) - 0x3FE6000000000000ull;
    const int64_t k = (int64_t)tmp >> 52;
    const uint64_t m = tmp & 0x000FFFFFFFFFFFFFull;
    const double z = __longlong_as_double((long long)(0x3FE6000000000000ull + m));
    const LogEntry e = tab[(uint32_t)(m >> 45)];
    const double r = fma(z, e.invc, -1.0);
    const double kd = (double)k;
    const double w = fma(kd, DST_LN2_HI, e.logc);
    double q = 1.0 / 9.0;
    q = fma(q, r, -1.0 / 8.0);
    q = fma(q, r, 1.0 / 7.0);
    q = fma(q, r, -1.0 / 6.0);
    q = fma(q, r, 1.0 / 5.0);
    q = fma(q, r, -1.0 / 4.0);
    q = fma(q, r, 1.0 / 3.0);
    q = fma(q, r, -1.0 / 2.0);
    const double lo = fma(kd, DST_LN2_LO, (r * r) * q);
    const double hi = w + r;
    return hi + ((w - hi) + r + lo);
}

// ---- finalisation math: tallies -> f64 in the reference's operation order ---------------------
// n / d for tallies below 2^24, correctly rounded like the IEEE division the reference executes, without the
// division's scaling / fix-up instructions and its f64 reciprocal: y = 1/d to ~2^-46 from the f32 reciprocal and
// one Newton step, then Markstein's correction q' = RN(q + (n - d q) y).  Why the rounding is right: n/d lies at least
// 1/(2d) > 2^-25 ulp away from every midpoint between adjacent doubles (the difference is a non-zero integer over
// 2d in units of an ulp), and q + (n - d q) y misses n/d by less than 2^-45 ulp.  d == 0 (no site where both records
// are known: NaN or inf) and larger tallies take the division.
// 1/b to ~2^-46 for an integer 0 < b < 2^24: the f32 reciprocal and one Newton step.  With it div_by(a, b, y) is
// RN(a/b) for every integer a < 2^24 (the argument above fin_raw) — not for other numerators.
__device__ __forceinline__ double rcp_int24(double b)
{
    const double y0 = (double)__builtin_amdgcn_rcpf((float)b);
    return fma(y0, fma(-b, y0, 1.0), y0);
}

// the IEEE divisions themselves, out of line: inlined beside the fast paths hipcc computes BOTH sides of the test and
// selects (a division has no side effect), which costs more than the division alone
__device__ __attribute__((noinline)) double div_plain(double a, double b) { return a / b; }

__device__ __forceinline__ double fin_raw(uint32_t n, uint32_t d)  // src/measures.rs:68
{
    const double nd = (double)n, dd = (double)d;
    if (d == 0 || (n | d) >> 24)
        return div_plain(nd, dd);
    const double y = rcp_int24(dd);
    const double q = nd * y;
    return fma(fma(-dd, q, nd), y, q);
}

__device__ __forceinline__ double fin_jc69(uint32_t n, uint32_t d, const LogEntry *tab = kLogTab)  // src/measures.rs:72-77
{
    const double p = fin_raw(n, d);
    return -0.75 * dst_log(1.0 - (4.0 / 3.0) * p, tab);
}

// Several correctly rounded quotients over ONE divisor: y = RN(1/b) by a real division, then per numerator
// Markstein's correction q = RN(a y), r = a - b q (exact in the fma), RN(q + r y) = RN(a/b) — three fused
// operations instead of a division's ~14, and the same bits (IEEE division is correctly rounded too).  It holds
// for finite non-zero b whose significand is not all ones, away from overflow / underflow: the callers check
// once that every divisor is a positive count or frequency sum (integers below 2^34 and their ratios) and
// send anything else (a record pair without known bases, ...) through the plain formulas.
__device__ __forceinline__ double div_by(double a, double b, double y)
{
    const double q = a * y;
    return fma(fma(-b, q, a), y, q);
}

__device__ __attribute__((noinline)) double fin_k80_plain(uint32_t count_L, uint32_t ts, uint32_t tv)
{
    const double P = (double)ts / (double)count_L;
    const double Q = (double)tv / (double)count_L;
    return -0.5 * dst_log((1.0 - 2.0 * P - Q) * sqrt(1.0 - 2.0 * Q));
}

__device__ __forceinline__ double fin_k80(uint32_t count_L, uint32_t ts, uint32_t tv, const LogEntry *tab = kLogTab)  // :109-112
{
    if (count_L == 0)
        return fin_k80_plain(count_L, ts, tv);
    // ts, tv <= count_L are integers: below 2^24 the quotients need no real division (rcp_int24)
    const double L = (double)count_L, inv_L = count_L >> 24 ? div_plain(1.0, L) : rcp_int24(L);
    const double P = div_by((double)ts, L, inv_L);
    const double Q = div_by((double)tv, L, inv_L);
    return -0.5 * dst_log((1.0 - 2.0 * P - Q) * sqrt(1.0 - 2.0 * Q), tab);
}

// counts = {A, T, G, C}; sums keep the reference's operand order (target first), :118-190
__device__ __attribute__((noinline)) double fin_tn93_plain(uint32_t count_L, uint32_t count_d, uint32_t count_P1,
                                                           uint32_t count_P2, uint4 qc, uint4 tc)
{
    const uint64_t L = (uint64_t)qc.x + qc.y + qc.z + qc.w + tc.x + tc.y + tc.z + tc.w;
    const double g_A = ((double)tc.x + (double)qc.x) / (double)L;
    const double g_C = ((double)tc.w + (double)qc.w) / (double)L;
    const double g_G = ((double)tc.z + (double)qc.z) / (double)L;
    const double g_T = ((double)tc.y + (double)qc.y) / (double)L;
    const double g_R = ((double)tc.x + (double)qc.x + (double)tc.z + (double)qc.z) / (double)L;
    const double g_Y = ((double)tc.w + (double)qc.w + (double)tc.y + (double)qc.y) / (double)L;
    const double k1 = 2.0 * g_A * g_G / g_R;
    const double k2 = 2.0 * g_T * g_C / g_Y;
    const double k3 = 2.0 * (g_R * g_Y - g_A * g_G * g_Y / g_R - g_T * g_C * g_R / g_Y);
    const double P1 = (double)count_P1 / (double)count_L;
    const double P2 = (double)count_P2 / (double)count_L;
    const double Q = (double)(count_d - (count_P1 + count_P2)) / (double)count_L;
    const double w1 = 1.0 - P1 / k1 - Q / (2.0 * g_R);
    const double w2 = 1.0 - P2 / k2 - Q / (2.0 * g_Y);
    const double w3 = 1.0 - Q / (2.0 * g_R * g_Y);
    double d = -k1 * dst_log(w1) - k2 * dst_log(w2) - k3 * dst_log(w3);
    if (d == 0.0)
        d = 0.0;
    return d;
}

// The same values with 5 divisions instead of 18: the quotients over L, count_L, g_R and g_Y share one
// reciprocal each (div_by), and the two integer divisors take theirs from rcp_int24.  The integer sums are exact in f64 (< 2^53), so adding them as integers first gives
// the reference's values.
__device__ __forceinline__ double fin_tn93(uint32_t count_L, uint32_t count_d, uint32_t count_P1,
                                           uint32_t count_P2, uint4 qc, uint4 tc, const LogEntry *tab = kLogTab)
{
    const uint64_t sA = (uint64_t)tc.x + qc.x, sT = (uint64_t)tc.y + qc.y, sG = (uint64_t)tc.z + qc.z,
                   sC = (uint64_t)tc.w + qc.w;
    if (count_L == 0 || sA + sG == 0 || sC + sT == 0)
        return fin_tn93_plain(count_L, count_d, count_P1, count_P2, qc, tc);
    // the base-frequency and P1 / P2 / Q quotients are integers over integers: below 2^24 they need no real division
    const uint64_t sL = sA + sT + sG + sC;
    const bool small = ((sL | (uint64_t)count_L | (uint64_t)count_d) >> 24) == 0;
    const double L = (double)sL, inv_L = small ? rcp_int24(L) : div_plain(1.0, L);
    const double g_A = div_by((double)sA, L, inv_L);
    const double g_C = div_by((double)sC, L, inv_L);
    const double g_G = div_by((double)sG, L, inv_L);
    const double g_T = div_by((double)sT, L, inv_L);
    const double g_R = div_by((double)(sA + sG), L, inv_L);
    const double g_Y = div_by((double)(sC + sT), L, inv_L);
    const double inv_R = 1.0 / g_R, inv_Y = 1.0 / g_Y;
    const double k1 = div_by(2.0 * g_A * g_G, g_R, inv_R);
    const double k2 = div_by(2.0 * g_T * g_C, g_Y, inv_Y);
    const double k3 = 2.0 * (g_R * g_Y - div_by(g_A * g_G * g_Y, g_R, inv_R) - div_by(g_T * g_C * g_R, g_Y, inv_Y));
    const double cL = (double)count_L, inv_cL = small ? rcp_int24(cL) : div_plain(1.0, cL);
    const double P1 = div_by((double)count_P1, cL, inv_cL);
    const double P2 = div_by((double)count_P2, cL, inv_cL);
    const double Q = div_by((double)(count_d - (count_P1 + count_P2)), cL, inv_cL);
    // Q / (2 g_R) = (Q / g_R) / 2: halving is exact
    const double w1 = 1.0 - P1 / k1 - 0.5 * div_by(Q, g_R, inv_R);
    const double w2 = 1.0 - P2 / k2 - 0.5 * div_by(Q, g_Y, inv_Y);
    const double w3 = 1.0 - Q / (2.0 * g_R * g_Y);
    double d = -k1 * dst_log(w1, tab) - k2 * dst_log(w2, tab) - k3 * dst_log(w3, tab);
    if (d == 0.0)
        d = 0.0;
    return d;
}

// ---- the fused epilogue's own arithmetic (DST_OUT_DISTANCE, dst_finalize_device): within 1e-12, not in the reference's order ----
// The pair kernels of jc69 / k80 / tn93 spend their time in this code (tn93: 234 f64 instructions per pair in the
// reference's operation order — five IEEE divisions, fourteen corrected quotients, three table logarithms), and nothing
// needs its last bits any more: the TSV text is finalised from the tallies by number_kernel with the functions above
// and the host's libm for the near ties (dst_text.hip), DST_OUT_TALLY + dst_finalize give the reference's bits.  So for
// the low-diversity alignments the consensus path exists for — every logarithm's argument within 2^-5 of 1 — the
// distance is evaluated as sums of -ln(1 - e) = e + e^2/2 + ... + e^9/9 (remainder below 3e-15 of the value) with
// reciprocals from the f32 unit and one Newton step (2^-44) and no quotient corrections: ~105 instructions for tn93,
// ~20 for jc69, ~32 for k80, absolute error below 1e-13.  Anything else — an argument further from 1, zero
// denominators, tallies of 2^24 and more, NaN — takes the functions above.
__device__ __forceinline__ double rcp_fast(double b)   // b > 0, inside the f32 range
{
    const double y0 = (double)__builtin_amdgcn_rcpf((float)b);
    return fma(y0, fma(-b, y0, 1.0), y0);
}

// -ln(1 - e) for 0 <= e < 2^-5
__device__ __forceinline__ double neg_ln1m(double e)
{
    double q = 1.0 / 9.0;
    q = fma(q, e, 1.0 / 8.0);
    q = fma(q, e, 1.0 / 7.0);
    q = fma(q, e, 1.0 / 6.0);
    q = fma(q, e, 1.0 / 5.0);
    q = fma(q, e, 1.0 / 4.0);
    q = fma(q, e, 1.0 / 3.0);
    q = fma(q, e, 1.0 / 2.0);
    q = fma(q, e, 1.0);
    return q * e;
}
constexpr double kSeriesMax = 0x1p-5;

__device__ __attribute__((noinline)) double fin_jc69_close(uint32_t n, uint32_t d, const LogEntry *tab) { return fin_jc69(n, d, tab); }
__device__ __attribute__((noinline)) double fin_k80_close(uint32_t count_L, uint32_t ts, uint32_t tv, const LogEntry *tab)
{
    return fin_k80(count_L, ts, tv, tab);
}
__device__ __attribute__((noinline)) double fin_tn93_close(uint32_t count_L, uint32_t count_d, uint32_t count_P1, uint32_t count_P2,
                                                           uint4 qc, uint4 tc, const LogEntry *tab)
{
    return fin_tn93(count_L, count_d, count_P1, count_P2, qc, tc, tab);
}

__device__ __forceinline__ double fin_jc69_fast(uint32_t n, uint32_t d, const LogEntry *tab)
{
    if (n == 0 && d != 0)
        return -0.0;   // -0.75 * ln(1): the reference's sign (src/measures.rs:76)
    const double e = (4.0 / 3.0) * ((double)n * rcp_fast((double)d));
    if (d == 0 || !(e < kSeriesMax))
        return fin_jc69_close(n, d, tab);
    return 0.75 * neg_ln1m(e);
}

__device__ __forceinline__ double fin_k80_fast(uint32_t count_L, uint32_t ts, uint32_t tv, const LogEntry *tab)
{
    if ((ts | tv) == 0 && count_L != 0)
        return -0.0;   // -0.5 * ln(1 * sqrt(1))
    const double inv_L = rcp_fast((double)count_L);
    const double P = (double)ts * inv_L, Q = (double)tv * inv_L;
    const double ea = 2.0 * P + Q, eb = 2.0 * Q;   // -0.5 ln((1 - ea) sqrt(1 - eb)) = 0.5 S(ea) + 0.25 S(eb)
    if (count_L == 0 || !(ea < kSeriesMax))         // (eb <= ea)
        return fin_k80_close(count_L, ts, tv, tab);
    return 0.5 * neg_ln1m(ea) + 0.25 * neg_ln1m(eb);
}

__device__ __forceinline__ double fin_tn93_fast(uint32_t count_L, uint32_t count_d, uint32_t count_P1, uint32_t count_P2, uint4 qc,
                                                uint4 tc, const LogEntry *tab)
{
    // Counts below 2^24 (alignments up to 16 M sites; larger ones go the long way): sums in 32 bits, one conversion
    // instruction each, and every product below stays inside the f32 exponent range of rcp_fast's seed.  Every product also
    // needs its factors non-zero.
    const uint32_t sA = tc.x + qc.x, sT = tc.y + qc.y, sG = tc.z + qc.z, sC = tc.w + qc.w;
    const uint32_t big = (qc.x | qc.y | qc.z | qc.w | tc.x | tc.y | tc.z | tc.w | count_L) >> 24;
    const uint32_t least = min(min(min(sA, sG), min(sT, sC)), count_L);
    if (big != 0 || least == 0)
        return fin_tn93_close(count_L, count_d, count_P1, count_P2, qc, tc, tab);
    const double A = (double)sA, G = (double)sG, T = (double)sT, C = (double)sC, cL = (double)count_L;
    const double R = A + G, Y = T + C, L = R + Y;   // (integers below 2^26: exact)
    const double AG = A * G, TC = T * C, RY = R * Y, AGTC = AG * TC;
    // six reciprocals from two (each a quarter-rate f32 seed and a Newton step): 1 / (R Y L) < 2^-78 and
    // 1 / (A G T C cL) < 2^-124 undone by the factors that are not wanted
    const double i1 = rcp_fast(RY * L), i2 = rcp_fast(AGTC * cL);
    const double iL = i1 * RY, iRY = i1 * L, iR = iRY * Y, iY = iRY * R;
    const double icL = i2 * AGTC, iAGTC = i2 * cL, iAG = iAGTC * TC, iTC = iAGTC * AG;
    // P1 / k1 = P1 L R / (2 A G);  Q / (2 g_R) = Q L / (2 R);  Q / (2 g_R g_Y) = Q L^2 / (2 R Y)      (g_X = X / L)
    const double u = icL * L;
    const double hq = 0.5 * (double)(count_d - (count_P1 + count_P2)) * u;
    const double e1 = fma(hq, iR, 0.5 * ((double)count_P1 * u) * (R * iAG));
    const double e2 = fma(hq, iY, 0.5 * ((double)count_P2 * u) * (Y * iTC));
    const double e3 = (hq * L) * iRY;
    if (!(e1 < kSeriesMax && e2 < kSeriesMax && e3 < kSeriesMax))
        return fin_tn93_close(count_L, count_d, count_P1, count_P2, qc, tc, tab);
    const double k1 = 2.0 * AG * (iL * iR), k2 = 2.0 * TC * (iL * iY);
    const double k3 = 2.0 * (iL * iL) * (RY - AG * (Y * iR) - TC * (R * iY));
    double d = k1 * neg_ln1m(e1) + k2 * neg_ln1m(e2) + k3 * neg_ln1m(e3);
    if (d == 0.0)
        d = 0.0;
    return d;
}

constexpr int OUT_TALLY = -1;      // uint32 x NT site tallies per pair
constexpr int OUT_INT = -2;        // int64 (n / n_high)
constexpr int OUT_TALLY_ADD = -3;  // split-L launch: atomicAdd partial tallies into a zeroed buffer
constexpr int OUT_INT_ADD = -4;    // split-L launch: atomicAdd partial counts into zeroed int64
constexpr int OUT_TALLY16 = -5;    // uint16 x NT per pair (alignments shorter than 65,536 sites)
// OUT >= 0: the measure id whose f64 distance the epilogue writes

// CLOSE: the reference's operation order with the table logarithm — within a few ulp of the host's libm (the text path,
// dst_finalize_device with DST_FIN_CLOSE); else the epilogue's arithmetic above (within 1e-12).  raw is the same either way.
template <int MEASURE, bool CLOSE = false>
__device__ __forceinline__ double finalize_pair(const uint32_t *o, uint4 qc, uint4 tc, const LogEntry *tab = kLogTab)
{
    if constexpr (MEASURE == DST_RAW)
        return fin_raw(o[0], o[1]);
    else if constexpr (MEASURE == DST_JC69)
        return CLOSE ? fin_jc69(o[0], o[1], tab) : fin_jc69_fast(o[0], o[1], tab);
    else if constexpr (MEASURE == DST_K80)
        return CLOSE ? fin_k80(o[0], o[1], o[2], tab) : fin_k80_fast(o[0], o[1], o[2], tab);
    else
        return CLOSE ? fin_tn93(o[0], o[1], o[2], o[3], qc, tc, tab) : fin_tn93_fast(o[0], o[1], o[2], o[3], qc, tc, tab);
}

// the same out of line: for kernels whose registers belong to their main loop (the dense pair kernels' sweep over L)
template <int MEASURE>
__device__ __attribute__((noinline)) double finalize_pair_call(uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, uint4 qc, uint4 tc)
{
    const uint32_t o[4] = {o0, o1, o2, o3};
    return finalize_pair<MEASURE>(o, qc, tc);
}

}  // namespace
}  // namespace dst
