// format.hpp — TSV number formatting of gather_write (src/lib.rs:626-633): Rust `{}` for i64 and
// `{:.12}` for f64.  Rust's fixed-precision Display prints the EXACT binary value rounded
// half-to-even at the 12th decimal, "NaN" / "inf" / "-inf", and keeps the sign of -0.0.
// fmt_fixed12 does that with 128-bit integer arithmetic (no libc printf in the hot loop);
// values >= 2^52 fall back to snprintf("%.12f"), which rounds the same way.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace cli {

constexpr int kFixed12Max = 336;  // sign + 309 digits + '.' + 12 digits + NUL, rounded up

// "00".."99"
inline const char *digit_pairs()
{
    static const char tab[201] = "00010203040506070809101112131415161718192021222324252627282930313233343536373839"
                                 "40414243444546474849505152535455565758596061626364656667686970717273747576777879"
                                 "8081828384858687888990919293949596979899";
    return tab;
}

inline int fmt_u64(uint64_t v, char *out)
{
    char tmp[24];
    int n = 0;
    const char *dp = digit_pairs();
    while (v >= 100) {
        const uint64_t q = v / 100;
        const unsigned r = (unsigned)(v - q * 100);
        tmp[n++] = dp[2 * r + 1];
        tmp[n++] = dp[2 * r];
        v = q;
    }
    if (v >= 10) {
        tmp[n++] = dp[2 * v + 1];
        tmp[n++] = dp[2 * v];
    } else {
        tmp[n++] = (char)('0' + v);
    }
    for (int k = 0; k < n; ++k)
        out[k] = tmp[n - 1 - k];
    return n;
}

inline int fmt_i64(int64_t v, char *out)
{
    if (v < 0) {
        out[0] = '-';
        return 1 + fmt_u64((uint64_t)(-(v + 1)) + 1, out + 1);
    }
    return fmt_u64((uint64_t)v, out);
}

// `out` must hold kFixed12Max bytes (DBL_MAX prints 309 integer digits); returns the length
inline int fmt_fixed12(double v, char *out)
{
    uint64_t bits;
    std::memcpy(&bits, &v, sizeof bits);
    const unsigned bexp = (unsigned)((bits >> 52) & 0x7FF);
    const uint64_t frac = bits & 0x000FFFFFFFFFFFFFull;
    int n = 0;
    if (bexp == 0x7FF) {
        if (frac) {
            std::memcpy(out, "NaN", 3);
            return 3;
        }
        if (bits >> 63)
            out[n++] = '-';
        std::memcpy(out + n, "inf", 3);
        return n + 3;
    }
    if (bits >> 63)
        out[n++] = '-';
    // |v| = m * 2^-sh exactly (normal: implicit leading one; subnormal: exponent of the smallest normal)
    const uint64_t m = bexp ? (frac | 0x0010000000000000ull) : frac;
    const int sh = 1075 - (int)(bexp ? bexp : 1);
    if (m != 0 && sh <= 0)                                  // >= 2^53: rare, let libc do it
        return n + std::snprintf(out + n, (size_t)(kFixed12Max - n), "%.12f", std::fabs(v));
    // R = round-half-even(m * 10^12 / 2^sh)
    unsigned __int128 R = 0;
    if (m != 0 && sh < 128) {
        const unsigned __int128 P = (unsigned __int128)m * 1000000000000ull;  // < 2^93
        R = P >> sh;
        const unsigned __int128 rem = P & ((((unsigned __int128)1) << sh) - 1);
        const unsigned __int128 half = ((unsigned __int128)1) << (sh - 1);
        if (rem > half || (rem == half && (R & 1)))
            R += 1;
    }
    uint64_t ip, fp;
    if ((uint64_t)(R >> 64) == 0) {                         // |v| < 1.8e7: everything a distance can be
        const uint64_t r64 = (uint64_t)R;
        ip = r64 / 1000000000000ull;
        fp = r64 - ip * 1000000000000ull;
    } else {
        ip = (uint64_t)(R / 1000000000000ull);
        fp = (uint64_t)(R % 1000000000000ull);
    }
    n += fmt_u64(ip, out + n);
    out[n++] = '.';
    const char *dp = digit_pairs();
    for (int k = 10; k >= 0; k -= 2) {
        const uint64_t q = fp / 100;
        const unsigned r = (unsigned)(fp - q * 100);
        out[n + k] = dp[2 * r];
        out[n + k + 1] = dp[2 * r + 1];
        fp = q;
    }
    return n + 12;
}

}  // namespace cli
