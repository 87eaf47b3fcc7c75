/*
 * distance_oracle.c — CPU restatement of the reference's hot path.  See distance_oracle.h:
 * test infrastructure only (checker + reported CPU baseline), never linked by the product.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  -ffp-contract=off keeps
 * the f64 operation order of src/measures.rs (rustc never fuses a*b+c); ln/sqrt are glibc's,
 * as in the reference binary.
 */
#include "distance_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- encoding.rs:4-41 ---- */
void orc_encoding_array(uint8_t a[256])
{
    static const struct { char c; uint8_t v; } letters[] = {
        {'A', 136}, {'G', 72},  {'C', 40},  {'T', 24},  {'R', 192}, {'M', 160}, {'W', 144},
        {'S', 96},  {'K', 80},  {'Y', 48},  {'V', 224}, {'H', 176}, {'D', 208}, {'B', 112},
        {'N', 240},
    };
    memset(a, 0, 256);
    for (size_t k = 0; k < sizeof letters / sizeof letters[0]; ++k) {
        a[(unsigned char)letters[k].c] = letters[k].v;             /* upper case */
        a[(unsigned char)(letters[k].c + ('a' - 'A'))] = letters[k].v; /* lower case */
    }
    a[(unsigned char)'-'] = 244;
    a[(unsigned char)'?'] = 242;
}

/* ---------------------------------------------------------------- fastaio.rs:101-118 -- */
size_t orc_encode(const uint8_t *chars, size_t len, uint8_t *codes)
{
    uint8_t table[256];
    orc_encoding_array(table); /* the reference rebuilds the table per call (fastaio.rs:102) */
    for (size_t i = 0; i < len; ++i) {
        if (table[chars[i]] == 0)
            return i + 1; /* Err(Invalid nucleotide character ...) fastaio.rs:111-113 */
        codes[i] = table[chars[i]];
    }
    return 0;
}

/* ---------------------------------------------------------------- fastaio.rs:120-145 -- */
size_t orc_encode_count_bases(const uint8_t *chars, size_t len, uint8_t *codes, uint64_t counts[4])
{
    uint8_t table[256];
    uint64_t counting[256];
    orc_encoding_array(table);
    memset(counting, 0, sizeof counting);
    for (size_t i = 0; i < len; ++i) {
        if (table[chars[i]] == 0)
            return i + 1;
        codes[i] = table[chars[i]];
        counting[chars[i]] += 1; /* raw characters, so lower case is NOT counted (:134) */
    }
    counts[0] = counting['A'];
    counts[1] = counting['T'];
    counts[2] = counting['G'];
    counts[3] = counting['C'];
    return 0;
}

/* ---------------------------------------------------------------- fastaio.rs:53-66 ---- */
void orc_count_bases(const uint8_t *codes, size_t len, uint64_t counts[4])
{
    uint64_t counting[256];
    memset(counting, 0, sizeof counting);
    for (size_t i = 0; i < len; ++i)
        counting[codes[i]] += 1;
    counts[0] = counting[136]; /* A */
    counts[1] = counting[24];  /* T */
    counts[2] = counting[72];  /* G */
    counts[3] = counting[40];  /* C */
}

/* ---------------------------------------------------------------- fastaio.rs:67-75 ---- */
size_t orc_get_differences(const uint8_t *codes, const uint8_t *other, size_t len, uint64_t *diffs)
{
    size_t k = 0;
    for (size_t i = 0; i < len; ++i)
        if (codes[i] < 240 && codes[i] != other[i])
            diffs[k++] = i;
    return k;
}

/* ---------------------------------------------------------------- fastaio.rs:289-336 -- */
static void consensus_count(const uint8_t *codes, size_t n, size_t len, size_t stride,
                            uint64_t *counts /* len x 4 */)
{
    /* lookup: 136->0 (A), 72->1 (G), 40->2 (C), 24->3 (T), everything else -> 0 (:295-302) */
    uint8_t lookup[256];
    memset(lookup, 0, sizeof lookup);
    lookup[72] = 1;
    lookup[40] = 2;
    lookup[24] = 3;
    for (size_t r = 0; r < n; ++r) {
        const uint8_t *row = codes + r * stride;
        for (size_t i = 0; i < len; ++i)
            counts[4 * i + lookup[row[i]]] += 1;
    }
}

static void consensus_pick(const uint64_t *counts, size_t len, uint8_t *cons)
{
    static const uint8_t back[4] = {136, 72, 40, 24}; /* A, G, C, T (:317-319) */
    for (size_t i = 0; i < len; ++i) {
        size_t maxidx = 0;
        uint64_t maxval = 0;
        for (size_t k = 0; k < 4; ++k) {
            if (counts[4 * i + k] > maxval) { /* strict >, so ties keep the earlier base */
                maxval = counts[4 * i + k];
                maxidx = k;
            }
        }
        cons[i] = back[maxidx];
    }
}

void orc_consensus(const uint8_t *codes, size_t n, size_t len, size_t stride, uint8_t *cons)
{
    uint64_t *counts = calloc(len ? 4 * len : 1, sizeof *counts);
    consensus_count(codes, n, len, stride, counts);
    consensus_pick(counts, len, cons);
    free(counts);
}

void orc_consensus2(const uint8_t *a, size_t na, size_t stride_a, const uint8_t *b, size_t nb,
                    size_t stride_b, size_t len, uint8_t *cons)
{
    uint64_t *counts = calloc(len ? 4 * len : 1, sizeof *counts);
    consensus_count(a, na, len, stride_a, counts);
    if (b)
        consensus_count(b, nb, len, stride_b, counts);
    consensus_pick(counts, len, cons);
    free(counts);
}

/* ---------------------------------------------------------------- measures.rs:14-23 --- */
int64_t orc_snp(const uint8_t *q, const uint8_t *t, size_t len)
{
    int64_t d = 0;
    for (size_t i = 0; i < len; ++i)
        if ((q[i] & t[i]) < 16)
            d += 1;
    return d;
}

/* ---------------------------------------------------------------- measures.rs:28-53 --- */
/* Rust's slice::binary_search: Ok(pos) when found.  The reference then sets `start = pos`,
 * a position RELATIVE to the slice it searched (:40-42); restated as-is. */
static int bsearch_u64(const uint64_t *v, size_t n, uint64_t key, size_t *pos)
{
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (v[mid] == key) {
            *pos = mid;
            return 1;
        }
        if (v[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return 0;
}

int64_t orc_snp_consensus(const uint8_t *q, const uint8_t *t, const uint64_t *q_diffs, size_t nq,
                          const uint64_t *t_diffs, size_t nt)
{
    int64_t d = 0;
    for (size_t k = 0; k < nq; ++k) {
        uint64_t idx = q_diffs[k];
        if ((q[idx] & t[idx]) < 16)
            d += 1;
    }
    size_t start = 0;
    for (size_t k = 0; k < nt; ++k) {
        uint64_t idx = t_diffs[k];
        size_t pos;
        if (bsearch_u64(q_diffs + start, nq - start, idx, &pos)) {
            start = pos;
            continue;
        }
        if ((q[idx] & t[idx]) < 16)
            d += 1;
    }
    return d;
}

/* ---------------------------------------------------------------- measures.rs:56-69 --- */
static void raw_tallies(const uint8_t *q, const uint8_t *t, size_t len, uint64_t *n_out,
                        uint64_t *d_out)
{
    uint64_t d = 0, n = 0;
    for (size_t i = 0; i < len; ++i) {
        if ((q[i] & 8) == 8 && q[i] == t[i]) {
            d += 1;
        } else if ((q[i] & t[i]) < 16) {
            d += 1;
            n += 1;
        }
    }
    *n_out = n;
    *d_out = d;
}

static double raw_final(uint64_t n, uint64_t d) { return (double)n / (double)d; }

double orc_raw(const uint8_t *q, const uint8_t *t, size_t len)
{
    uint64_t n, d;
    raw_tallies(q, t, len, &n, &d);
    return raw_final(n, d);
}

/* ---------------------------------------------------------------- measures.rs:72-77 --- */
static double jc69_final(double p) { return -0.75 * log(1.0 - (4.0 / 3.0) * p); }

double orc_jc69(const uint8_t *q, const uint8_t *t, size_t len)
{
    return jc69_final(orc_raw(q, t, len));
}

/* ---------------------------------------------------------------- measures.rs:80-113 -- */
static void k80_tallies(const uint8_t *q, const uint8_t *t, size_t len, uint64_t *L_out,
                        uint64_t *ts_out, uint64_t *tv_out)
{
    uint64_t count_L = 0, ts = 0, tv = 0;
    for (size_t i = 0; i < len; ++i) {
        uint8_t a = q[i], b = t[i];
        if ((a & 8) == 8 && a == b) {
            count_L += 1;
        } else if ((a & b) < 16) {
            if ((a & 55) == 0 && (b & 55) == 0) { /* both purines */
                ts += 1;
                count_L += 1;
            } else if ((a & 199) == 0 && (b & 199) == 0) { /* both pyrimidines */
                ts += 1;
                count_L += 1;
            } else if (((a & 55) == 0 && (b & 199) == 0) || ((a & 199) == 0 && (b & 55) == 0)) {
                tv += 1;
                count_L += 1;
            }
        }
    }
    *L_out = count_L;
    *ts_out = ts;
    *tv_out = tv;
}

static double k80_final(uint64_t count_L, uint64_t ts, uint64_t tv)
{
    double P = (double)ts / (double)count_L;
    double Q = (double)tv / (double)count_L;
    return -0.5 * log((1.0 - 2.0 * P - Q) * sqrt(1.0 - 2.0 * Q));
}

double orc_k80(const uint8_t *q, const uint8_t *t, size_t len)
{
    uint64_t L, ts, tv;
    k80_tallies(q, t, len, &L, &ts, &tv);
    return k80_final(L, ts, tv);
}

/* ---------------------------------------------------------------- measures.rs:116-193 - */
static void tn93_tallies(const uint8_t *q, const uint8_t *t, size_t len, uint64_t *L_out,
                         uint64_t *d_out, uint64_t *p1_out, uint64_t *p2_out)
{
    uint64_t count_P1 = 0, count_P2 = 0, count_d = 0, count_L = 0;
    for (size_t i = 0; i < len; ++i) {
        uint8_t a = q[i], b = t[i];
        if ((a & 8) == 8 && a == b) {
            count_L += 1;
        } else if ((a & b) < 16 && (a & 8) == 8 && (b & 8) == 8) {
            count_d += 1;
            count_L += 1;
            if ((a | b) == 200)
                count_P1 += 1;
            else if ((a | b) == 56)
                count_P2 += 1;
        }
    }
    *L_out = count_L;
    *d_out = count_d;
    *p1_out = count_P1;
    *p2_out = count_P2;
}

/* counts = {A, T, G, C}.  Operand order of every sum follows :118-143 (target first). */
static double tn93_final(uint64_t count_L, uint64_t count_d, uint64_t count_P1, uint64_t count_P2,
                         const uint64_t qc[4], const uint64_t tc[4])
{
    const uint64_t qA = qc[0], qT = qc[1], qG = qc[2], qC = qc[3];
    const uint64_t tA = tc[0], tT = tc[1], tG = tc[2], tC = tc[3];
    uint64_t L = qA + qT + qG + qC + tA + tT + tG + tC;

    double g_A = ((double)tA + (double)qA) / (double)L;
    double g_C = ((double)tC + (double)qC) / (double)L;
    double g_G = ((double)tG + (double)qG) / (double)L;
    double g_T = ((double)tT + (double)qT) / (double)L;
    double g_R = ((double)tA + (double)qA + (double)tG + (double)qG) / (double)L;
    double g_Y = ((double)tC + (double)qC + (double)tT + (double)qT) / (double)L;

    double k1 = 2.0 * g_A * g_G / g_R;
    double k2 = 2.0 * g_T * g_C / g_Y;
    double k3 = 2.0 * (g_R * g_Y - g_A * g_G * g_Y / g_R - g_T * g_C * g_R / g_Y);

    double P1 = (double)count_P1 / (double)count_L;
    double P2 = (double)count_P2 / (double)count_L;
    double Q = (double)(count_d - (count_P1 + count_P2)) / (double)count_L;

    double w1 = 1.0 - P1 / k1 - Q / (2.0 * g_R);
    double w2 = 1.0 - P2 / k2 - Q / (2.0 * g_Y);
    double w3 = 1.0 - Q / (2.0 * g_R * g_Y);

    double d = -k1 * log(w1) - k2 * log(w2) - k3 * log(w3);
    if (d == 0.0) /* normalises -0.0 to +0.0 (:188-190) */
        d = 0.0;
    return d;
}

double orc_tn93(const uint8_t *q, const uint8_t *t, size_t len, const uint64_t q_counts[4],
                const uint64_t t_counts[4])
{
    uint64_t L, d, p1, p2;
    tn93_tallies(q, t, len, &L, &d, &p1, &p2);
    return tn93_final(L, d, p1, p2, q_counts, t_counts);
}

/* ---------------------------------------------------------------- tallies / finalise -- */
int orc_tallies(int measure, const uint8_t *q, const uint8_t *t, size_t len, uint64_t out[4])
{
    switch (measure) {
    case ORC_N:
    case ORC_N_HIGH:
        out[0] = (uint64_t)orc_snp(q, t, len);
        return 1;
    case ORC_RAW:
    case ORC_JC69:
        raw_tallies(q, t, len, &out[0], &out[1]);
        return 2;
    case ORC_K80:
        k80_tallies(q, t, len, &out[0], &out[1], &out[2]);
        return 3;
    case ORC_TN93:
        tn93_tallies(q, t, len, &out[0], &out[1], &out[2], &out[3]);
        return 4;
    default:
        return -1;
    }
}

double orc_finalize(int measure, const uint64_t tl[4], const uint64_t q_counts[4],
                    const uint64_t t_counts[4])
{
    switch (measure) {
    case ORC_N:
    case ORC_N_HIGH:
        return (double)tl[0];
    case ORC_RAW:
        return raw_final(tl[0], tl[1]);
    case ORC_JC69:
        return jc69_final(raw_final(tl[0], tl[1]));
    case ORC_K80:
        return k80_final(tl[0], tl[1], tl[2]);
    case ORC_TN93:
        return tn93_final(tl[0], tl[1], tl[2], tl[3], q_counts, t_counts);
    default:
        return NAN;
    }
}

/* ---------------------------------------------------------------- lib.rs:502-596 ------ */
size_t orc_pairs_square(size_t n, uint64_t *ij)
{
    size_t k = 0;
    if (n == 0)
        return 0; /* the reference's `0..n-1` underflows for n == 0; load_fasta rejects it first */
    for (size_t i = 0; i < n - 1; ++i)
        for (size_t j = i + 1; j < n; ++j) {
            ij[2 * k] = i;
            ij[2 * k + 1] = j;
            ++k;
        }
    return k;
}

size_t orc_pairs_rectangle(size_t n1, size_t n2, uint64_t *ij)
{
    size_t k = 0;
    for (size_t i = 0; i < n1; ++i)
        for (size_t j = 0; j < n2; ++j) {
            ij[2 * k] = i;
            ij[2 * k + 1] = j;
            ++k;
        }
    return k;
}

/* ---------------------------------------------------------------- lib.rs:626-633 ------ */
int orc_format_int(int64_t v, char *buf, size_t cap) { return snprintf(buf, cap, "%lld", (long long)v); }

int orc_format_float(double v, char *buf, size_t cap)
{
    /* Rust `{:.12}`: NaN -> "NaN" (no sign), +-inf -> "inf"/"-inf", -0.0 keeps its sign,
     * finite values are exactly rounded like glibc's %.12f. */
    if (isnan(v))
        return snprintf(buf, cap, "NaN");
    if (isinf(v))
        return snprintf(buf, cap, v < 0 ? "-inf" : "inf");
    return snprintf(buf, cap, "%.12f", v);
}

/* ---------------------------------------------------------------- all-pairs drivers --- */
typedef struct {
    int measure;
    const uint8_t *a, *b; /* b == a for square */
    size_t na, nb, len, stride_a, stride_b;
    const uint64_t *counts_a, *counts_b;  /* n x 4 */
    uint64_t *const *diffs_a, *const *diffs_b;
    const size_t *ndiffs_a, *ndiffs_b;
    int square;
    uint64_t begin, end; /* canonical pair-index range of this worker */
    double *out;         /* indexed by pair index - out_base */
    uint64_t out_base;
} job_t;

static double one_pair(const job_t *jb, size_t i, size_t j)
{
    const uint8_t *q = jb->a + i * jb->stride_a; /* record_1, lib.rs:432 */
    const uint8_t *t = jb->b + j * jb->stride_b; /* record_2, lib.rs:433 */
    switch (jb->measure) {
    case ORC_N:
        return (double)orc_snp_consensus(q, t, jb->diffs_a[i], jb->ndiffs_a[i], jb->diffs_b[j],
                                         jb->ndiffs_b[j]);
    case ORC_N_HIGH:
        return (double)orc_snp(q, t, jb->len);
    case ORC_RAW:
        return orc_raw(q, t, jb->len);
    case ORC_JC69:
        return orc_jc69(q, t, jb->len);
    case ORC_K80:
        return orc_k80(q, t, jb->len);
    case ORC_TN93:
        return orc_tn93(q, t, jb->len, jb->counts_a + 4 * i, jb->counts_b + 4 * j);
    default:
        return NAN;
    }
}

/* first canonical index of row i in the i<j triangle of n */
static uint64_t tri_row_start(uint64_t n, uint64_t i) { return i * (2 * n - i - 1) / 2; }

static void *worker(void *arg)
{
    const job_t *jb = arg;
    if (jb->begin >= jb->end)
        return NULL;
    if (jb->square) {
        /* locate (i, j) of pair index `begin`, then walk row-major like generate_pairs_square */
        uint64_t n = jb->na, i = 0;
        while (i + 1 < n && tri_row_start(n, i + 1) <= jb->begin)
            ++i;
        uint64_t j = i + 1 + (jb->begin - tri_row_start(n, i));
        for (uint64_t p = jb->begin; p < jb->end; ++p) {
            jb->out[p - jb->out_base] = one_pair(jb, i, j);
            if (++j == n) {
                ++i;
                j = i + 1;
            }
        }
    } else {
        for (uint64_t p = jb->begin; p < jb->end; ++p)
            jb->out[p - jb->out_base] = one_pair(jb, p / jb->nb, p % jb->nb);
    }
    return NULL;
}

static int run_jobs(job_t *proto, uint64_t begin, uint64_t end, int threads)
{
    if (threads < 1)
        threads = 1; /* lib.rs:255-256 */
    pthread_t *tid = calloc((size_t)threads, sizeof *tid);
    job_t *jobs = calloc((size_t)threads, sizeof *jobs);
    uint64_t total = end - begin;
    for (int k = 0; k < threads; ++k) {
        jobs[k] = *proto;
        jobs[k].begin = begin + total * (uint64_t)k / (uint64_t)threads;
        jobs[k].end = begin + total * (uint64_t)(k + 1) / (uint64_t)threads;
        jobs[k].out_base = begin;
    }
    for (int k = 1; k < threads; ++k)
        pthread_create(&tid[k], NULL, worker, &jobs[k]);
    worker(&jobs[0]);
    for (int k = 1; k < threads; ++k)
        pthread_join(tid[k], NULL);
    free(tid);
    free(jobs);
    return 0;
}

typedef struct {
    uint64_t *counts;
    uint64_t **diffs;
    size_t *ndiffs;
} prep_t;

/* per-measure precompute of set_up(), lib.rs:219-242 */
static void prep_set(int measure, const uint8_t *codes, size_t n, size_t len, size_t stride,
                     const uint64_t *given_counts, const uint8_t *cons, prep_t *p)
{
    memset(p, 0, sizeof *p);
    if (measure == ORC_TN93 && !given_counts) {
        p->counts = calloc(n ? 4 * n : 1, sizeof *p->counts);
        for (size_t r = 0; r < n; ++r)
            orc_count_bases(codes + r * stride, len, p->counts + 4 * r);
    }
    if (measure == ORC_N) {
        p->diffs = calloc(n ? n : 1, sizeof *p->diffs);
        p->ndiffs = calloc(n ? n : 1, sizeof *p->ndiffs);
        uint64_t *tmp = malloc((len ? len : 1) * sizeof *tmp);
        for (size_t r = 0; r < n; ++r) {
            size_t k = orc_get_differences(codes + r * stride, cons, len, tmp);
            p->diffs[r] = malloc((k ? k : 1) * sizeof(uint64_t));
            memcpy(p->diffs[r], tmp, k * sizeof(uint64_t));
            p->ndiffs[r] = k;
        }
        free(tmp);
    }
}

static void prep_free(prep_t *p, size_t n)
{
    free(p->counts);
    if (p->diffs)
        for (size_t r = 0; r < n; ++r)
            free(p->diffs[r]);
    free(p->diffs);
    free(p->ndiffs);
}

int orc_all_pairs_square(int measure, const uint8_t *codes, size_t n, size_t len, size_t stride,
                         const uint64_t *counts, uint64_t pair_begin, uint64_t pair_end,
                         int threads, double *out)
{
    if (measure < ORC_N || measure > ORC_TN93)
        return -1;
    uint64_t total = n ? (uint64_t)n * (n - 1) / 2 : 0;
    if (pair_end > total)
        pair_end = total;
    if (pair_begin >= pair_end)
        return 0;
    uint8_t *cons = NULL;
    if (measure == ORC_N) {
        cons = malloc(len ? len : 1);
        orc_consensus(codes, n, len, stride, cons);
    }
    prep_t p;
    prep_set(measure, codes, n, len, stride, counts, cons, &p);
    job_t jb = {0};
    jb.measure = measure;
    jb.a = jb.b = codes;
    jb.na = jb.nb = n;
    jb.len = len;
    jb.stride_a = jb.stride_b = stride;
    jb.counts_a = jb.counts_b = counts ? counts : p.counts;
    jb.diffs_a = jb.diffs_b = p.diffs;
    jb.ndiffs_a = jb.ndiffs_b = p.ndiffs;
    jb.square = 1;
    jb.out = out;
    run_jobs(&jb, pair_begin, pair_end, threads);
    prep_free(&p, n);
    free(cons);
    return 0;
}

int orc_all_pairs_rect(int measure, const uint8_t *a, size_t na, size_t stride_a,
                       const uint64_t *counts_a, const uint8_t *b, size_t nb, size_t stride_b,
                       const uint64_t *counts_b, size_t len, int threads, double *out)
{
    if (measure < ORC_N || measure > ORC_TN93)
        return -1;
    uint8_t *cons = NULL;
    if (measure == ORC_N) {
        cons = malloc(len ? len : 1);
        orc_consensus2(a, na, stride_a, b, nb, stride_b, len, cons); /* lib.rs:223 */
    }
    prep_t pa, pb;
    prep_set(measure, a, na, len, stride_a, counts_a, cons, &pa);
    prep_set(measure, b, nb, len, stride_b, counts_b, cons, &pb);
    job_t jb = {0};
    jb.measure = measure;
    jb.a = a;
    jb.b = b;
    jb.na = na;
    jb.nb = nb;
    jb.len = len;
    jb.stride_a = stride_a;
    jb.stride_b = stride_b;
    jb.counts_a = counts_a ? counts_a : pa.counts;
    jb.counts_b = counts_b ? counts_b : pb.counts;
    jb.diffs_a = pa.diffs;
    jb.diffs_b = pb.diffs;
    jb.ndiffs_a = pa.ndiffs;
    jb.ndiffs_b = pb.ndiffs;
    jb.square = 0;
    jb.out = out;
    run_jobs(&jb, 0, (uint64_t)na * nb, threads);
    prep_free(&pa, na);
    prep_free(&pb, nb);
    free(cons);
    return 0;
}

/* ---------------------------------------------------------------- slabs: tallies -> values -> TSV ---- */
/* The tallies of rows [rb, re) of a square job (canonical order, `width` uint32 per pair: what the engine's
 * DST_OUT_TALLY gives) finalised like measures.rs does (orc_finalize), and gather_write's lines for such a slab
 * (lib.rs:626-633); rows shared out over `threads`. */
typedef struct {
    int measure, width;
    const uint32_t *tallies;
    const uint64_t *counts;
    size_t n;
    uint64_t rb, r0, r1; /* this worker's rows [r0, r1) of the slab that starts at row rb */
    double *out;
    int is_int;
    const double *values;
    const char *id_chars;
    const uint64_t *id_offs;
    char *text;
    uint64_t text_at;
} slab_job_t;

static void *finalize_rows(void *arg)
{
    const slab_job_t *jb = arg;
    const uint64_t base = tri_row_start(jb->n, jb->rb);
    for (uint64_t i = jb->r0; i < jb->r1; ++i) {
        uint64_t p = tri_row_start(jb->n, i) - base;
        for (uint64_t j = i + 1; j < jb->n; ++j, ++p) {
            uint64_t tl[4] = {0, 0, 0, 0};
            for (int k = 0; k < jb->width; ++k)
                tl[k] = jb->tallies[p * (uint64_t)jb->width + (uint64_t)k];
            jb->out[p] = orc_finalize(jb->measure, tl, jb->counts ? jb->counts + 4 * i : NULL,
                                      jb->counts ? jb->counts + 4 * j : NULL);
        }
    }
    return NULL;
}

/* equal pair counts, not equal row counts: the triangle's rows shrink */
static void split_rows(slab_job_t *jobs, const slab_job_t *proto, uint64_t rb, uint64_t re, int threads)
{
    const uint64_t base = tri_row_start(proto->n, rb), total = tri_row_start(proto->n, re) - base;
    uint64_t r = rb;
    for (int k = 0; k < threads; ++k) {
        jobs[k] = *proto;
        jobs[k].r0 = r;
        const uint64_t target = total * (uint64_t)(k + 1) / (uint64_t)threads;
        while (r < re && (k == threads - 1 || tri_row_start(proto->n, r) - base < target))
            ++r;
        jobs[k].r1 = r;
    }
}

static void run_slab_jobs(slab_job_t *jobs, int threads, void *(*fn)(void *))
{
    pthread_t *tid = calloc((size_t)threads, sizeof *tid);
    for (int k = 1; k < threads; ++k)
        pthread_create(&tid[k], NULL, fn, &jobs[k]);
    fn(&jobs[0]);
    for (int k = 1; k < threads; ++k)
        pthread_join(tid[k], NULL);
    free(tid);
}

int orc_finalize_square(int measure, const uint32_t *tallies, int width, size_t n, const uint64_t *counts,
                        uint64_t rb, uint64_t re, int threads, double *out)
{
    if (measure < ORC_N || measure > ORC_TN93 || width < 1 || width > 4 || re > n || rb > re)
        return -1;
    if (threads < 1)
        threads = 1;
    slab_job_t proto = {0};
    proto.measure = measure;
    proto.width = width;
    proto.tallies = tallies;
    proto.counts = counts;
    proto.n = n;
    proto.rb = rb;
    proto.out = out;
    slab_job_t *jobs = calloc((size_t)threads, sizeof *jobs);
    split_rows(jobs, &proto, rb, re, threads);
    run_slab_jobs(jobs, threads, finalize_rows);
    free(jobs);
    return 0;
}

/* the lines of rows [r0, r1); text == NULL only measures them */
static uint64_t tsv_rows(const slab_job_t *jb, char *text)
{
    const uint64_t base = tri_row_start(jb->n, jb->rb);
    uint64_t at = 0;
    char num[64];
    for (uint64_t i = jb->r0; i < jb->r1; ++i) {
        const char *id1 = jb->id_chars + jb->id_offs[i];
        const size_t l1 = (size_t)(jb->id_offs[i + 1] - jb->id_offs[i]);
        uint64_t p = tri_row_start(jb->n, i) - base;
        for (uint64_t j = i + 1; j < jb->n; ++j, ++p) {
            const size_t l2 = (size_t)(jb->id_offs[j + 1] - jb->id_offs[j]);
            const int ln = jb->is_int ? orc_format_int((int64_t)jb->values[p], num, sizeof num)
                                      : orc_format_float(jb->values[p], num, sizeof num);
            if (text) {
                char *o = text + at;
                memcpy(o, id1, l1);
                o[l1] = '\t';
                memcpy(o + l1 + 1, jb->id_chars + jb->id_offs[j], l2);
                o[l1 + 1 + l2] = '\t';
                memcpy(o + l1 + l2 + 2, num, (size_t)ln);
                o[l1 + l2 + 2 + (size_t)ln] = '\n';
            }
            at += l1 + l2 + (uint64_t)ln + 3;
        }
    }
    return at;
}

static void *tsv_measure(void *arg)
{
    slab_job_t *jb = arg;
    jb->text_at = tsv_rows(jb, NULL);
    return NULL;
}

static void *tsv_write(void *arg)
{
    slab_job_t *jb = arg;
    tsv_rows(jb, jb->text + jb->text_at);
    return NULL;
}

/* returns the length of the text; nothing is written when it exceeds cap */
uint64_t orc_tsv_square(int is_int, const double *values, size_t n, uint64_t rb, uint64_t re, const char *id_chars,
                        const uint64_t *id_offs, char *out, uint64_t cap, int threads)
{
    if (re > n || rb > re)
        return 0;
    if (threads < 1)
        threads = 1;
    slab_job_t proto = {0};
    proto.is_int = is_int;
    proto.values = values;
    proto.n = n;
    proto.rb = rb;
    proto.id_chars = id_chars;
    proto.id_offs = id_offs;
    proto.text = out;
    slab_job_t *jobs = calloc((size_t)threads, sizeof *jobs);
    split_rows(jobs, &proto, rb, re, threads);
    run_slab_jobs(jobs, threads, tsv_measure);
    uint64_t total = 0;
    for (int k = 0; k < threads; ++k) {
        const uint64_t len = jobs[k].text_at;
        jobs[k].text_at = total;
        total += len;
    }
    if (out && total <= cap)
        run_slab_jobs(jobs, threads, tsv_write);
    free(jobs);
    return total;
}
