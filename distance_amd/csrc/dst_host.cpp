// dst_host.cpp — host-side logic of the engine that needs no GPU: canonical pair order and its
// partition (src/lib.rs:502-596), tile scheduling, finalisation in the reference's f64 operation
// order (src/measures.rs) and the TSV number format (src/lib.rs:626-633).
// Built with -ffp-contract=off: rustc never contracts a*b+c, and ln/sqrt are glibc's in both.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "dst_internal.h"

namespace dst {

int tally_width(int measure)
{
    switch (measure) {
    case DST_N:
    case DST_N_HIGH: return 1;
    case DST_RAW:
    case DST_JC69: return 2;
    case DST_K80: return 3;
    case DST_TN93: return 4;
    default: return 0;
    }
}

bool measure_is_int(int measure) { return measure == DST_N || measure == DST_N_HIGH; }

// index of pair (i, i+1) in the i<j row-major enumeration of generate_pairs_square (lib.rs:511-512)
uint64_t square_row_start(uint64_t n, uint64_t i) { return i * (2 * n - i - 1) / 2; }

uint64_t pairs_in_rows(bool square, uint64_t n_cols, uint64_t row_begin, uint64_t row_end)
{
    if (row_end <= row_begin)
        return 0;
    if (!square)
        return (row_end - row_begin) * n_cols;
    const uint64_t e = std::min(row_end, n_cols);
    const uint64_t b = std::min(row_begin, e);
    return square_row_start(n_cols, e) - square_row_start(n_cols, b);
}

// Tiles of one launch.  A column tile ("panel") of BN records is shared by every row tile that
// meets it; the panel is the big operand (BN >> BM), so all row tiles of one panel are queued
// back to back on ONE of 8 queues, and the queues are interleaved so that block ids b, b+8, ...
// (observed to land on one XCD: MI355X_MICROARCH.md "Workgroup dispatch") drain one queue.
// Placement only affects speed (L2 hits on the panel), never results.
std::vector<BlockDesc> build_blocks(bool square, uint64_t row_begin, uint64_t row_end,
                                    uint64_t n_cols, TileShape ts)
{
    constexpr int kQueues = 8;
    std::vector<BlockDesc> out;
    if (row_end <= row_begin || n_cols == 0)
        return out;
    const uint64_t BM = (uint64_t)ts.bm, BN = (uint64_t)ts.bn;
    const uint64_t n_panels = (n_cols + BN - 1) / BN;
    // measurement knob (tools/kbench.py): DST_SCHEDULE=rowmajor deals tiles row tile by row tile
    // with no regard for XCDs, to price the panel-per-XCD schedule below against it
    const char *mode = std::getenv("DST_SCHEDULE");
    if (mode && std::strcmp(mode, "rowmajor") == 0) {
        for (uint64_t i0 = row_begin; i0 < row_end; i0 += BM)
            for (uint64_t pj = 0; pj < n_panels; ++pj) {
                const uint64_t jmax = std::min(n_cols, pj * BN + BN) - 1;
                if (!square || i0 < jmax)
                    out.push_back({(uint32_t)i0, (uint32_t)(pj * BN)});
            }
        return out;
    }

    struct Panel {
        uint64_t j0, first_row, n_tiles;
    };
    std::vector<Panel> panels;
    for (uint64_t pj = 0; pj < n_panels; ++pj) {
        const uint64_t j0 = pj * BN;
        const uint64_t jmax = std::min(n_cols, j0 + BN) - 1;  // last real column of the panel
        // rows i that pair with some column of the panel: square needs i < jmax
        const uint64_t r_end = square ? std::min(row_end, jmax) : row_end;
        if (r_end <= row_begin)
            continue;
        panels.push_back({j0, row_begin, (r_end - row_begin + BM - 1) / BM});
    }
    // longest panels first onto the currently shortest queue
    std::vector<size_t> order(panels.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(),
                     [&](size_t a, size_t b) { return panels[a].n_tiles > panels[b].n_tiles; });
    std::vector<std::vector<BlockDesc>> queue(kQueues);
    // a panel much longer than a fair share is split over several queues (few-panel launches)
    uint64_t total_tiles = 0;
    for (const Panel &p : panels)
        total_tiles += p.n_tiles;
    const uint64_t fair = std::max<uint64_t>(1, (total_tiles + kQueues - 1) / kQueues);
    for (size_t idx : order) {
        const Panel &p = panels[idx];
        uint64_t done = 0;
        while (done < p.n_tiles) {
            size_t q = 0;
            for (size_t k = 1; k < (size_t)kQueues; ++k)
                if (queue[k].size() < queue[q].size())
                    q = k;
            const uint64_t room = fair > queue[q].size() ? fair - queue[q].size() : 1;
            const uint64_t take = std::min(p.n_tiles - done, std::max<uint64_t>(room, 1));
            for (uint64_t t = 0; t < take; ++t)
                queue[q].push_back({(uint32_t)(p.first_row + (done + t) * BM), (uint32_t)p.j0});
            done += take;
        }
    }
    size_t longest = 0;
    for (const auto &q : queue)
        longest = std::max(longest, q.size());
    out.reserve(longest * kQueues);
    for (size_t k = 0; k < longest; ++k)
        for (int q = 0; q < kQueues; ++q)
            out.push_back(k < queue[q].size() ? queue[q][k] : BlockDesc{0xFFFFFFFFu, 0u});
    while (!out.empty() && out.back().i0 == 0xFFFFFFFFu)
        out.pop_back();
    return out;
}

// ---- shared preparation: which records a rank prepares, and the shape of its exchange block ------------
// every rank takes rmax records (a multiple of 256: whole waves of the pack), the last ranks possibly fewer or none
SharedLayout shared_layout(uint64_t n, int world, uint32_t ent_cap)
{
    SharedLayout lay{};
    lay.world = (uint32_t)world;
    const uint64_t per = (n + (uint64_t)world - 1) / (uint64_t)world;
    lay.rmax = (uint32_t)((per + 255) / 256 * 256);
    lay.cnt_at = 16;
    lay.counts_at = lay.cnt_at + lay.rmax;
    lay.ent_at = lay.counts_at + 4 * lay.rmax;
    lay.ent_cap = (ent_cap + 3u) & ~3u;
    lay.words = lay.ent_at + lay.ent_cap;
    return lay;
}

// ---- consensus-delta path: per-site semantics, tables, tiles ---------------------------------------

int family_of(int measure)
{
    switch (measure) {
    case DST_N:
    case DST_N_HIGH: return FAM_NHIGH;
    case DST_RAW:
    case DST_JC69: return FAM_RAW;
    case DST_K80: return FAM_K80;
    default: return FAM_TN93;
    }
}

int family_words(int family, bool wide)
{
    const int nt = family == FAM_NHIGH ? 1 : family == FAM_RAW ? 2 : family == FAM_K80 ? 3 : 4;
    return wide ? nt : (nt + 1) / 2;
}

// What ONE site adds to each tally: the bodies of the site loops of src/measures.rs.
void site_tallies(int measure, uint8_t q, uint8_t t, int out[4])
{
    out[0] = out[1] = out[2] = out[3] = 0;
    const bool same = (q & 8) == 8 && q == t;  // "are the bases certainly the same"
    const bool differ = (q & t) < 16;          // "they are certainly different"
    switch (measure) {
    case DST_N:
    case DST_N_HIGH:  // src/measures.rs:14-23 (and :28-53, which counts the same sites)
        out[0] = differ;
        break;
    case DST_RAW:
    case DST_JC69:  // src/measures.rs:59-66 -> {n, d}
        if (same) {
            out[1] = 1;
        } else if (differ) {
            out[0] = 1;
            out[1] = 1;
        }
        break;
    case DST_K80:  // src/measures.rs:85-107 -> {count_L, ts, tv}
        if (same) {
            out[0] = 1;
        } else if (differ) {
            const bool q_pur = (q & 55) == 0, t_pur = (t & 55) == 0, q_pyr = (q & 199) == 0, t_pyr = (t & 199) == 0;
            if ((q_pur && t_pur) || (q_pyr && t_pyr)) {
                out[1] = 1;
                out[0] = 1;
            } else if ((q_pur && t_pyr) || (q_pyr && t_pur)) {
                out[2] = 1;
                out[0] = 1;
            }
        }
        break;
    case DST_TN93:  // src/measures.rs:156-175 -> {count_L, count_d, count_P1, count_P2}
        if (same) {
            out[0] = 1;
        } else if (differ && (q & 8) == 8 && (t & 8) == 8) {
            out[1] = 1;
            out[0] = 1;
            if ((q | t) == 200)
                out[2] = 1;
            else if ((q | t) == 56)
                out[3] = 1;
        }
        break;
    default: break;
    }
}

namespace {

// the code src/encoding.rs gives a set of bases (high nibble): known bases carry bit 3, the N class is N itself
uint8_t code_of_nibble(int nib)
{
    const bool one = nib == 8 || nib == 4 || nib == 2 || nib == 1;
    return (uint8_t)(nib == 15 ? 240 : (nib << 4) | (one ? 8 : 0));
}

void pack_words(int family, bool wide, const int64_t d[4], uint32_t w[kMaxWords])
{
    const int nt = family == FAM_NHIGH ? 1 : family == FAM_RAW ? 2 : family == FAM_K80 ? 3 : 4;
    for (int k = 0; k < kMaxWords; ++k)
        w[k] = 0;
    if (wide || nt == 1) {
        for (int k = 0; k < nt; ++k)
            w[k] = (uint32_t)d[k];
        return;
    }
    // two tallies per word, exact modulo 2^32: word = low + 65536 * high
    w[0] = (uint32_t)(d[0] + d[1] * 65536);
    if (nt == 3)
        w[1] = (uint32_t)d[2];
    else if (nt == 4)
        w[1] = (uint32_t)(d[2] + d[3] * 65536);
}

}  // namespace

void pack_tallies(int family, bool wide, const int64_t d[4], uint32_t w[kMaxWords]) { pack_words(family, wide, d, w); }

void build_consensus_lut(ConsensusLut &lut)
{
    static const int kFamilyMeasure[4] = {DST_N_HIGH, DST_RAW, DST_K80, DST_TN93};
    static const int kRefNibble[kRefClasses] = {8, 4, 2, 1, 15};
    std::memset(&lut, 0, sizeof lut);
    for (int fam = 0; fam < 4; ++fam)
        for (int wide = 0; wide < 2; ++wide) {
            const int m = kFamilyMeasure[fam];
            int kk[4];
            site_tallies(m, 136, 136, kk);  // a known base against itself (the same for all four)
            int64_t u[4] = {kk[0], kk[1], kk[2], kk[3]};
            pack_words(fam, wide != 0, u, lut.unit[fam][wide]);
            for (int cls = 0; cls < kRefClasses; ++cls) {
                const uint8_t c = code_of_nibble(kRefNibble[cls]);
                int fcc[4];
                site_tallies(m, c, c, fcc);
                for (int x = 1; x < 16; ++x) {
                    const uint8_t qx = code_of_nibble(x);
                    int fxc[4];
                    site_tallies(m, qx, c, fxc);
                    int64_t a[4];
                    for (int k = 0; k < 4; ++k)
                        a[k] = (int64_t)fxc[k] - fcc[k];
                    pack_words(fam, wide != 0, a, lut.a[fam][wide][cls][x]);
                    for (int y = 1; y < 16; ++y) {
                        const uint8_t ty = code_of_nibble(y);
                        int fxy[4], fcy[4];
                        site_tallies(m, qx, ty, fxy);
                        site_tallies(m, c, ty, fcy);
                        int64_t h[4];
                        for (int k = 0; k < 4; ++k)
                            h[k] = (int64_t)fxy[k] - fxc[k] - fcy[k] + fcc[k];
                        pack_words(fam, wide != 0, h, lut.h[fam][wide][cls][x][y]);
                    }
                }
            }
        }
}

// Tiles of one consensus-path launch: panel-major, so the workgroups in flight share a panel's site
// buckets; rows that cannot pair with any column of a panel (square: i >= its last column) are left out.
std::vector<ConsensusTile> build_consensus_tiles(bool square, uint64_t row_begin, uint64_t row_end,
                                                 uint64_t n_cols, uint32_t rows_per_tile)
{
    std::vector<ConsensusTile> out;
    if (row_end <= row_begin || n_cols == 0 || rows_per_tile == 0)
        return out;
    const uint64_t n_panels = (n_cols + kPanelCols - 1) / kPanelCols;
    for (uint64_t p = 0; p < n_panels; ++p) {
        const uint64_t last = std::min<uint64_t>(n_cols, (p + 1) * kPanelCols) - 1;  // last column of the panel
        const uint64_t r_end = square ? std::min(row_end, last) : row_end;
        for (uint64_t i0 = row_begin; i0 < r_end; i0 += rows_per_tile)
            out.push_back({(uint32_t)i0, (uint32_t)std::min<uint64_t>(r_end, i0 + rows_per_tile), (uint32_t)p});
    }
    return out;
}

}  // namespace dst

// ================================================================================================
// C ABI: host-only entry points
// ================================================================================================
using namespace dst;

extern "C" {

int dst_abi_version(void) { return DST_ABI_VERSION; }

int dst_measure_from_name(const char *name)
{
    if (!name)
        return -1;
    static const char *names[] = {"n", "n_high", "raw", "jc69", "k80", "tn93"};
    for (int k = 0; k < 6; ++k)
        if (std::strcmp(name, names[k]) == 0)
            return k;
    return -1;
}

int dst_tally_width(int measure) { return tally_width(measure); }

int dst_site_tallies(int measure, uint8_t q, uint8_t t, int *out)
{
    if (measure < DST_N || measure > DST_TN93 || !out)
        return DST_ERR_ARG;
    int o[4];
    site_tallies(measure, q, t, o);
    for (int k = 0; k < tally_width(measure); ++k)
        out[k] = o[k];
    return DST_OK;
}

const char *dst_status_string(int status)
{
    switch (status) {
    case DST_OK: return "ok";
    case DST_ERR_ARG: return "bad argument";
    case DST_ERR_HIP: return "HIP runtime error";
    case DST_ERR_INVALID_CODE: return "invalid nucleotide code";
    case DST_ERR_STATE: return "bad state";
    case DST_ERR_NOMEM: return "out of memory";
    case DST_ERR_CAPACITY: return "output buffer too small";
    default: return "unknown status";
    }
}

uint64_t dst_square_pairs(uint64_t n) { return n ? n * (n - 1) / 2 : 0; }

uint64_t dst_square_row_start(uint64_t n, uint64_t i) { return square_row_start(n, i); }

int dst_partition_square(uint64_t n, int parts, uint64_t *bounds)
{
    if (parts < 1 || !bounds)
        return DST_ERR_ARG;
    const uint64_t total = dst_square_pairs(n);
    bounds[0] = 0;
    uint64_t row = 0;
    for (int k = 1; k < parts; ++k) {
        // first row whose start index reaches k/parts of the pairs (128-bit safe for n < 2^32)
        const uint64_t target = (uint64_t)(((unsigned __int128)total * (unsigned)k) / (unsigned)parts);
        uint64_t lo = row, hi = n;
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2;
            if (square_row_start(n, mid) < target)
                lo = mid + 1;
            else
                hi = mid;
        }
        row = lo;
        bounds[k] = row;
    }
    bounds[parts] = n;
    return DST_OK;
}

int dst_partition_rect(uint64_t n_rows, int parts, uint64_t *bounds)
{
    if (parts < 1 || !bounds)
        return DST_ERR_ARG;
    for (int k = 0; k <= parts; ++k)
        bounds[k] = (uint64_t)(((unsigned __int128)n_rows * (unsigned)k) / (unsigned)parts);
    return DST_OK;
}

int dst_shared_range(uint64_t n, int rank, int world, uint64_t *begin, uint64_t *end)
{
    if (world < 1 || rank < 0 || rank >= world || !begin || !end)
        return DST_ERR_ARG;
    const SharedLayout lay = shared_layout(n, world, 0);
    *begin = std::min<uint64_t>(n, (uint64_t)rank * lay.rmax);
    *end = std::min<uint64_t>(n, ((uint64_t)rank + 1) * lay.rmax);
    return DST_OK;
}

int dst_shared_block_layout(uint64_t n, int world, uint32_t entries, uint32_t layout[6])
{
    if (world < 1 || !layout)
        return DST_ERR_ARG;
    const SharedLayout lay = shared_layout(n, world, entries);
    layout[0] = lay.rmax;
    layout[1] = lay.cnt_at;
    layout[2] = lay.counts_at;
    layout[3] = lay.ent_at;
    layout[4] = lay.ent_cap;
    layout[5] = lay.words;
    return DST_OK;
}

size_t dst_out_bytes(int measure, int out_kind, uint64_t n_pairs)
{
    if (out_kind == DST_OUT_TALLY)
        return (size_t)n_pairs * (size_t)tally_width(measure) * sizeof(uint32_t);
    if (out_kind == DST_OUT_TALLY16)
        return (size_t)n_pairs * (size_t)tally_width(measure) * sizeof(uint16_t);
    return (size_t)n_pairs * 8;
}

// ---- finalisation: src/measures.rs, same expressions, same order -------------------------------
static double fin_raw(uint64_t n, uint64_t d) { return (double)n / (double)d; }  // :68

static double fin_jc69(double p) { return -0.75 * std::log(1.0 - (4.0 / 3.0) * p); }  // :76

static double fin_k80(uint64_t count_L, uint64_t ts, uint64_t tv)  // :109-112
{
    const double P = (double)ts / (double)count_L;
    const double Q = (double)tv / (double)count_L;
    return -0.5 * std::log((1.0 - 2.0 * P - Q) * std::sqrt(1.0 - 2.0 * Q));
}

static double fin_tn93(const uint32_t *tl, const uint32_t *qc, const uint32_t *tc)  // :118-190
{
    const uint64_t qA = qc[0], qT = qc[1], qG = qc[2], qC = qc[3];
    const uint64_t tA = tc[0], tT = tc[1], tG = tc[2], tC = tc[3];
    const uint64_t L = qA + qT + qG + qC + tA + tT + tG + tC;
    const double g_A = ((double)tA + (double)qA) / (double)L;
    const double g_C = ((double)tC + (double)qC) / (double)L;
    const double g_G = ((double)tG + (double)qG) / (double)L;
    const double g_T = ((double)tT + (double)qT) / (double)L;
    const double g_R = ((double)tA + (double)qA + (double)tG + (double)qG) / (double)L;
    const double g_Y = ((double)tC + (double)qC + (double)tT + (double)qT) / (double)L;
    const double k1 = 2.0 * g_A * g_G / g_R;
    const double k2 = 2.0 * g_T * g_C / g_Y;
    const double k3 = 2.0 * (g_R * g_Y - g_A * g_G * g_Y / g_R - g_T * g_C * g_R / g_Y);
    const uint64_t count_L = tl[0], count_d = tl[1], count_P1 = tl[2], count_P2 = tl[3];
    const double P1 = (double)count_P1 / (double)count_L;
    const double P2 = (double)count_P2 / (double)count_L;
    const double Q = (double)(count_d - (count_P1 + count_P2)) / (double)count_L;
    const double w1 = 1.0 - P1 / k1 - Q / (2.0 * g_R);
    const double w2 = 1.0 - P2 / k2 - Q / (2.0 * g_Y);
    const double w3 = 1.0 - Q / (2.0 * g_R * g_Y);
    double d = -k1 * std::log(w1) - k2 * std::log(w2) - k3 * std::log(w3);
    if (d == 0.0)
        d = 0.0;
    return d;
}

int dst_finalize(int measure, const uint32_t *tallies, const uint32_t *q_counts,
                 const uint32_t *t_counts, double *as_float, int64_t *as_int)
{
    if (!tallies)
        return DST_ERR_ARG;
    switch (measure) {
    case DST_N:
    case DST_N_HIGH:
        if (!as_int)
            return DST_ERR_ARG;
        *as_int = (int64_t)tallies[0];
        return DST_OK;
    case DST_RAW:
    case DST_JC69:
    case DST_K80:
    case DST_TN93: break;
    default: return DST_ERR_ARG;
    }
    if (!as_float)
        return DST_ERR_ARG;
    if (measure == DST_RAW)
        *as_float = fin_raw(tallies[0], tallies[1]);
    else if (measure == DST_JC69)
        *as_float = fin_jc69(fin_raw(tallies[0], tallies[1]));
    else if (measure == DST_K80)
        *as_float = fin_k80(tallies[0], tallies[1], tallies[2]);
    else {
        if (!q_counts || !t_counts)
            return DST_ERR_ARG;
        *as_float = fin_tn93(tallies, q_counts, t_counts);
    }
    return DST_OK;
}

// Rust `{}` for i64 and `{:.12}` for f64 (src/lib.rs:626-633)
int dst_format_distance(int measure, double as_float, int64_t as_int, char *buf, size_t cap)
{
    if (!buf || cap == 0)
        return -1;
    if (measure_is_int(measure))
        return std::snprintf(buf, cap, "%lld", (long long)as_int);
    if (std::isnan(as_float))
        return std::snprintf(buf, cap, "NaN");
    if (std::isinf(as_float))
        return std::snprintf(buf, cap, as_float < 0 ? "-inf" : "inf");
    return std::snprintf(buf, cap, "%.12f", as_float);
}

}  // extern "C"
