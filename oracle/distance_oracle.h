/*
 * distance_oracle.h — CPU restatement of benjamincjackson/distance's all-pairs hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / reported CPU baseline.  The shipped path (distance_amd/, include/) never
 * links, loads or falls back to it.
 *
 * Parity pinning: the Rust reference cannot be built here (no cargo/rustc, crates not
 * vendored), so this restatement is pinned by the reference's own unit-test vectors
 * (tests/golden/reference_vectors.json, each entry cites the reference file:line) plus
 * exhaustive code-pair truth tables derived from the source semantics.  Results outside those
 * vectors (ambiguity codes, NaN/inf/-0 printing, FASTA corner cases) are "parity unpinned"
 * by the reference and pinned only by this restatement of its source.
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 */
#ifndef DISTANCE_ORACLE_H
#define DISTANCE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* measure ids shared with include/distance_hip.h (names = the CLI's -m values, lib.rs:104-109) */
enum {
    ORC_N = 0,      /* snp_consensus  measures.rs:28-53  */
    ORC_N_HIGH = 1, /* snp            measures.rs:14-23  */
    ORC_RAW = 2,    /* raw            measures.rs:56-69  */
    ORC_JC69 = 3,   /* jc69           measures.rs:72-77  */
    ORC_K80 = 4,    /* k80            measures.rs:80-113 */
    ORC_TN93 = 5    /* tn93           measures.rs:116-193 */
};

/* src/encoding.rs:4-41 — 256-entry char -> Paradis code table (0 = invalid). */
void orc_encoding_array(uint8_t table[256]);

/* src/fastaio.rs:101-118 `encode`: returns 0, or 1 + index of the first invalid character. */
size_t orc_encode(const uint8_t *chars, size_t len, uint8_t *codes);

/* src/fastaio.rs:120-145 `encode_count_bases`: as orc_encode, and counts RAW uppercase
 * 'A','T','G','C' characters only (fastaio.rs:136-142).  counts = {A, T, G, C}. */
size_t orc_encode_count_bases(const uint8_t *chars, size_t len, uint8_t *codes, uint64_t counts[4]);

/* src/fastaio.rs:53-66 `count_bases`: histogram of codes 136/24/72/40.  counts = {A, T, G, C}. */
void orc_count_bases(const uint8_t *codes, size_t len, uint64_t counts[4]);

/* src/fastaio.rs:67-75 `get_differences`: sites with seq[i] < 240 && seq[i] != other[i].
 * Writes at most len indices into diffs, returns how many. */
size_t orc_get_differences(const uint8_t *codes, const uint8_t *other, size_t len, uint64_t *diffs);

/* src/fastaio.rs:289-336 `consensus` over n rows of a row-major n x len matrix (stride bytes). */
void orc_consensus(const uint8_t *codes, size_t n, size_t len, size_t stride, uint8_t *cons);
/* same, continuing the column counts over a second set (the reference walks every loaded file). */
void orc_consensus2(const uint8_t *a, size_t na, size_t stride_a, const uint8_t *b, size_t nb,
                    size_t stride_b, size_t len, uint8_t *cons);

/* src/measures.rs — the six per-pair functions.  q = query, t = target. */
int64_t orc_snp(const uint8_t *q, const uint8_t *t, size_t len);                       /* :14-23  */
int64_t orc_snp_consensus(const uint8_t *q, const uint8_t *t, const uint64_t *q_diffs,
                          size_t nq, const uint64_t *t_diffs, size_t nt);              /* :28-53  */
double orc_raw(const uint8_t *q, const uint8_t *t, size_t len);                        /* :56-69  */
double orc_jc69(const uint8_t *q, const uint8_t *t, size_t len);                       /* :72-77  */
double orc_k80(const uint8_t *q, const uint8_t *t, size_t len);                        /* :80-113 */
double orc_tn93(const uint8_t *q, const uint8_t *t, size_t len, const uint64_t q_counts[4],
                const uint64_t t_counts[4]);                                           /* :116-193 */

/* Integer site tallies of the same loops (what the GPU path emits before finalisation).
 * raw/jc69: out = {n, d}; k80: {count_L, ts, tv}; tn93: {count_L, count_d, count_P1, count_P2};
 * n/n_high: {d}.  Returns the number of tallies written. */
int orc_tallies(int measure, const uint8_t *q, const uint8_t *t, size_t len, uint64_t out[4]);

/* Finalisation alone, in the reference's f64 operation order, from integer tallies. */
double orc_finalize(int measure, const uint64_t tallies[4], const uint64_t q_counts[4],
                    const uint64_t t_counts[4]);

/* src/lib.rs:502-549 / 551-596 — canonical pair order.  Fill (i, j) for every pair. */
size_t orc_pairs_square(size_t n, uint64_t *ij);
size_t orc_pairs_rectangle(size_t n1, size_t n2, uint64_t *ij);

/* src/lib.rs:626-633 — one distance as the TSV writer prints it: `{}` for Int, `{:.12}` for
 * Float (Rust Display: "NaN", "inf", "-inf", "-0.000000000000").  Returns chars written. */
int orc_format_int(int64_t v, char *buf, size_t cap);
int orc_format_float(double v, char *buf, size_t cap);

/*
 * All-pairs drivers with the shape of lib.rs:413-458 (T worker threads over contiguous pair
 * ranges; no channels / clones / formatting, i.e. an upper bound on the reference binary).
 * codes: row-major n x stride.  out: canonical order (square: i<j row-major; rect: i outer).
 * For ints (n, n_high) the f64 holds the exact integer.  counts: n x 4 {A,T,G,C} or NULL
 * (tn93 then counts bases itself, as lib.rs:233-239).  pair_begin/pair_end select a
 * sub-range of canonical pair indices (for bounded timing samples).
 */
int orc_all_pairs_square(int measure, const uint8_t *codes, size_t n, size_t len, size_t stride,
                         const uint64_t *counts, uint64_t pair_begin, uint64_t pair_end,
                         int threads, double *out);
int orc_all_pairs_rect(int measure, const uint8_t *a, size_t na, size_t stride_a,
                       const uint64_t *counts_a, const uint8_t *b, size_t nb, size_t stride_b,
                       const uint64_t *counts_b, size_t len, int threads, double *out);

/*
 * Slab checkers for the engine's TSV text: the tallies of rows [rb, re) of a square job (canonical order,
 * `width` uint32 per pair) finalised with orc_finalize (measures.rs:68, 76, 109-112, 118-190; libm's log),
 * and gather_write's lines for such a slab (lib.rs:626-633: "id1\tid2\tvalue\n", `{}` / `{:.12}`).
 * orc_tsv_square returns the text's length and writes it only when it fits cap.
 */
int orc_finalize_square(int measure, const uint32_t *tallies, int width, size_t n, const uint64_t *counts,
                        uint64_t rb, uint64_t re, int threads, double *out);
uint64_t orc_tsv_square(int is_int, const double *values, size_t n, uint64_t rb, uint64_t re, const char *id_chars,
                        const uint64_t *id_offs, char *out, uint64_t cap, int threads);

#ifdef __cplusplus
}
#endif
#endif
