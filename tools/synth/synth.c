/*
 * synth.c — the synthetic alignments of SURVEY.md §8(d), for bench.py and the full-size GPU tests.
 * Bench / test infrastructure, not part of the product library.
 *
 *   deterministic: xoshiro256** seeded (through splitmix64) with  seed ^ stream id
 *   root:     L sites, P(A,C,G,T) = (0.30, 0.18, 0.20, 0.32)
 *   record r: root with Poisson(L/1000) substitutions at uniform sites (uniform over the 3 other
 *             bases); every site becomes N w.p. 1e-3 and a random 2-/3-fold IUPAC code w.p. 1e-4;
 *             1 % of records get a leading and a trailing '-' run of U[0,200] sites each.
 *   Codes are emitted directly (src/encoding.rs values); to_fasta() writes letters, 5 % of the
 *   records in lower case.
 * Every record has its own generator (seeded with the record index), so any range of records —
 * a streamed batch, a host slice for the CPU baseline — is generated independently and is the
 * same bytes whoever generates it.
 */
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    uint64_t s[4];
} rng_t;

static uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static void rng_seed(rng_t *r, uint64_t seed)
{
    for (int k = 0; k < 4; ++k)
        r->s[k] = splitmix64(&seed);
}

static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static inline uint64_t rng_next(rng_t *r)  /* xoshiro256** */
{
    uint64_t *s = r->s;
    const uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return result;
}

static inline double rng_unit(rng_t *r) { return (double)(rng_next(r) >> 11) * 0x1.0p-53; }          /* [0,1) */
static inline double rng_open(rng_t *r) { return ((double)(rng_next(r) >> 11) + 0.5) * 0x1.0p-53; }  /* (0,1) */
static inline uint64_t rng_below(rng_t *r, uint64_t n) { return (uint64_t)(rng_unit(r) * (double)n); }

static const uint8_t kBase[4] = {136, 40, 72, 24};                                     /* A C G T */
static const uint8_t kIupac[10] = {192, 160, 144, 96, 80, 48, 224, 176, 208, 112};     /* R M W S K Y V H D B */

void synth_root(uint64_t seed, size_t len, uint8_t *root)
{
    rng_t r;
    rng_seed(&r, seed ^ 0xA5A5A5A5A5A5A5A5ull);
    for (size_t i = 0; i < len; ++i) {
        const double u = rng_unit(&r);
        root[i] = kBase[u < 0.30 ? 0 : u < 0.48 ? 1 : u < 0.68 ? 2 : 3];
    }
}

static void one_record(uint64_t seed, const uint8_t *root, size_t len, uint64_t index, uint8_t *row)
{
    rng_t r;
    rng_seed(&r, seed ^ (0x9E3779B97F4A7C15ull * (index + 1)));
    memcpy(row, root, len);
    if (len == 0)
        return;
    /* Poisson(len/1000) substitutions: count the unit-rate arrivals before time lambda */
    const double lambda = (double)len / 1000.0;
    for (double t = -log(rng_open(&r)); t <= lambda; t += -log(rng_open(&r))) {
        const size_t site = (size_t)rng_below(&r, len);
        int b = 0;
        while (kBase[b] != root[site])
            ++b;
        row[site] = kBase[(b + 1 + (int)rng_below(&r, 3)) & 3];
    }
    /* N w.p. 1e-3, IUPAC w.p. 1e-4 per site: geometric gaps between the affected sites */
    const double p = 1.1e-3, lq = log(1.0 - p);
    for (double at = floor(log(rng_open(&r)) / lq); at < (double)len; at += 1.0 + floor(log(rng_open(&r)) / lq)) {
        const size_t site = (size_t)at;
        row[site] = rng_unit(&r) < (1.0 / 11.0) ? kIupac[rng_below(&r, 10)] : 240;
    }
    if (rng_unit(&r) < 0.01) {
        size_t lead = (size_t)rng_below(&r, 201), trail = (size_t)rng_below(&r, 201);
        if (lead > len)
            lead = len;
        if (trail > len)
            trail = len;
        memset(row, 244, lead);
        memset(row + len - trail, 244, trail);
    }
}

typedef struct {
    uint64_t seed;
    const uint8_t *root;
    size_t len, first, n, stride;
    uint8_t *out;
    int tid, threads;
} job_t;

static void *worker(void *arg)
{
    const job_t *j = (const job_t *)arg;
    const size_t lo = j->n * (size_t)j->tid / (size_t)j->threads, hi = j->n * (size_t)(j->tid + 1) / (size_t)j->threads;
    for (size_t k = lo; k < hi; ++k)
        one_record(j->seed, j->root, j->len, j->first + k, j->out + k * j->stride);
    return NULL;
}

/* records [first, first + n) of the alignment (seed, root) into out (rows `stride` bytes apart) */
void synth_records(uint64_t seed, const uint8_t *root, size_t len, size_t first, size_t n, uint8_t *out,
                   size_t stride, int threads)
{
    if (threads < 1)
        threads = 1;
    if (threads > 64)
        threads = 64;
    pthread_t th[64];
    job_t jobs[64];
    for (int t = 0; t < threads; ++t) {
        jobs[t] = (job_t){seed, root, len, first, n, stride, out, t, threads};
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t)
        pthread_join(th[t], NULL);
}

/* codes -> FASTA letters; record `index` is written in lower case when its generator says so (5 %) */
void synth_letters(uint64_t seed, const uint8_t *row, size_t len, uint64_t index, char *out)
{
    static const char up[] = "AGCTRMWSKYVHDBN-?";
    static const uint8_t codes[] = {136, 72, 40, 24, 192, 160, 144, 96, 80, 48, 224, 176, 208, 112, 240, 244, 242};
    char lut[256];
    memset(lut, '?', sizeof lut);
    uint64_t x = seed ^ (0xD1B54A32D192ED03ull * (index + 1));
    const int lower = (splitmix64(&x) % 100) < 5;
    for (int k = 0; k < 17; ++k)
        lut[codes[k]] = (char)((lower && up[k] >= 'A' && up[k] <= 'Z') ? up[k] - 'A' + 'a' : up[k]);
    for (size_t i = 0; i < len; ++i)
        out[i] = lut[row[i]];
}
