// host_check.cpp — sanitizer driver (ASan + UBSan, CPU build only) for the host-only logic in
// distance_amd/csrc/dst_host.cpp: tile schedules, partitions, finalisation, number formatting.
// Built and run by tests/test_host_sanitizers.py; exits non-zero on any failed check.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#include "../../distance_amd/csrc/dst_internal.h"
#include "../../distance_amd/cli/format.hpp"

using namespace dst;

static int failures = 0;
#define CHECK(c)                                                       \
    do {                                                               \
        if (!(c)) {                                                    \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            ++failures;                                                \
        }                                                              \
    } while (0)

static void check_schedule(bool square, uint64_t rb, uint64_t re, uint64_t ncols, TileShape ts)
{
    const std::vector<BlockDesc> blocks = build_blocks(square, rb, re, ncols, ts);
    std::map<std::pair<uint32_t, uint32_t>, int> seen;
    uint64_t covered = 0;
    for (const BlockDesc &b : blocks) {
        if (b.i0 == 0xFFFFFFFFu)
            continue;
        CHECK(b.i0 >= rb && b.i0 < re && b.j0 % (uint32_t)ts.bn == 0);
        const int dup = seen[std::make_pair(b.i0, b.j0)]++;
        CHECK(dup == 0);
        for (uint64_t i = b.i0; i < std::min<uint64_t>(b.i0 + ts.bm, re); ++i)
            for (uint64_t j = b.j0; j < std::min<uint64_t>(b.j0 + ts.bn, ncols); ++j)
                covered += (!square || j > i) ? 1 : 0;
    }
    CHECK(covered == pairs_in_rows(square, ncols, rb, re));
}

int main()
{
    for (TileShape ts : {TileShape{12, 512}, TileShape{24, 512}, TileShape{8, 512}, TileShape{32, 512}})
        for (uint64_t n : {1ull, 2ull, 11ull, 512ull, 513ull, 1500ull, 4097ull}) {
            check_schedule(true, 0, n, n, ts);
            check_schedule(true, n / 3, n - n / 4, n, ts);
            check_schedule(false, 0, n, 700, ts);
            check_schedule(false, n / 2, n, 1, ts);
        }
    for (uint64_t n : {0ull, 1ull, 2ull, 3ull, 100ull, 50000ull, 4000000000ull})
        for (int parts : {1, 2, 3, 8, 64}) {
            std::vector<uint64_t> b((size_t)parts + 1);
            CHECK(dst_partition_square(n, parts, b.data()) == DST_OK);
            CHECK(b[0] == 0 && b[(size_t)parts] == n);
            for (int k = 0; k < parts; ++k)
                CHECK(b[(size_t)k] <= b[(size_t)k + 1]);
            CHECK(dst_partition_rect(n, parts, b.data()) == DST_OK && b[(size_t)parts] == n);
        }
    CHECK(dst_partition_square(10, 0, nullptr) == DST_ERR_ARG);
    // finalisation incl. degenerate tallies (0/0, ln of negatives, empty counts)
    const uint32_t zero4[4] = {0, 0, 0, 0}, cnt[4] = {5, 6, 7, 8};
    double f = 0;
    int64_t iv = 0;
    for (int m = DST_N; m <= DST_TN93; ++m) {
        const uint32_t t1[4] = {10, 3, 1, 1};
        CHECK(dst_finalize(m, t1, cnt, cnt, &f, &iv) == DST_OK);
        CHECK(dst_finalize(m, zero4, zero4, zero4, &f, &iv) == DST_OK);
        CHECK(dst_finalize(m, nullptr, cnt, cnt, &f, &iv) == DST_ERR_ARG);
    }
    CHECK(dst_finalize(DST_TN93, zero4, nullptr, cnt, &f, &iv) == DST_ERR_ARG);
    CHECK(dst_finalize(DST_RAW, zero4, nullptr, nullptr, &f, &iv) == DST_OK && std::isnan(f));
    // formatting into exact-size buffers
    char buf[cli::kFixed12Max];
    for (double v : {0.0, -0.0, 2.0 / 15.0, 1e300, -1.7976931348623157e308, 4.9e-324, 123456789.123456789,
                     (double)NAN, (double)INFINITY}) {
        const int n = dst_format_distance(DST_RAW, v, 0, buf, sizeof buf);
        CHECK(n > 0 && n < (int)sizeof buf);
        char fast[cli::kFixed12Max];   // exactly the documented size: ASan catches an overrun
        const int k = cli::fmt_fixed12(v, fast);
        CHECK(k == n && std::memcmp(fast, buf, (size_t)n) == 0);
    }
    char tiny[8];                      // snprintf semantics: returns the needed length, never overruns
    CHECK(dst_format_distance(DST_RAW, 0.5, 0, tiny, sizeof tiny) == 14 && tiny[7] == 0);
    CHECK(dst_format_distance(DST_N, 0.0, -9223372036854775807LL - 1, buf, sizeof buf) == 20);
    CHECK(cli::fmt_i64(-9223372036854775807LL - 1, buf) == 20 && std::memcmp(buf, "-9223372036854775808", 20) == 0);
    CHECK(dst_format_distance(DST_RAW, 1.0, 0, nullptr, 0) == -1);
    CHECK(dst_measure_from_name(nullptr) == -1 && dst_measure_from_name("tn93") == DST_TN93);
    if (failures == 0)
        std::puts("host_check: all checks passed");
    return failures ? 1 : 0;
}
