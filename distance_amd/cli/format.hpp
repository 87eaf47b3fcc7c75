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

inline int fmt_u64(uint64_t v, char *out)
{
    char tmp[24];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
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
    if (std::isnan(v)) {
        std::memcpy(out, "NaN", 3);
        return 3;
    }
    int n = 0;
    if (std::signbit(v))
        out[n++] = '-';
    if (std::isinf(v)) {
        std::memcpy(out + n, "inf", 3);
        return n + 3;
    }
    const double a = std::fabs(v);
    int e2;
    const double fr = std::frexp(a, &e2);                   // a = fr * 2^e2, fr in [0.5, 1)
    const uint64_t m = (uint64_t)std::ldexp(fr, 53);        // exact 53-bit integer (0 for a == 0)
    const int sh = 53 - e2;                                 // a = m * 2^-sh
    if (a != 0.0 && sh <= 0)                                // >= 2^53: rare, let libc do it
        return n + std::snprintf(out + n, (size_t)(kFixed12Max - n), "%.12f", a);
    unsigned __int128 R = 0;
    if (a != 0.0 && sh < 128) {
        const unsigned __int128 P = (unsigned __int128)m * 1000000000000ull;  // < 2^93
        R = P >> sh;
        const unsigned __int128 rem = P & ((((unsigned __int128)1) << sh) - 1);
        const unsigned __int128 half = ((unsigned __int128)1) << (sh - 1);
        if (rem > half || (rem == half && (R & 1)))
            R += 1;
    }
    const uint64_t ip = (uint64_t)(R / 1000000000000ull);
    uint64_t fp = (uint64_t)(R % 1000000000000ull);
    n += fmt_u64(ip, out + n);
    out[n++] = '.';
    for (int k = 11; k >= 0; --k) {
        out[n + k] = (char)('0' + fp % 10);
        fp /= 10;
    }
    return n + 12;
}

}  // namespace cli
