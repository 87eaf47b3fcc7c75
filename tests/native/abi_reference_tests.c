/* abi_reference_tests.c — the reference's own unit tests (src/measures.rs:195-309,
 * src/lib.rs:906-1154), re-stated in plain C against the C ABI of libdistance_hip.so: what a
 * compiled host (the Rust reference via extern "C") would run.
 * Built with gcc and run on the GPU box by tests/test_gpu_abi_c.py.  Exit code = failures. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/distance_hip.h"

static int failures = 0;
#define CHECK(c)                                                            \
    do {                                                                    \
        if (!(c)) {                                                         \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            ++failures;                                                     \
        }                                                                   \
    } while (0)
#define OK(call) CHECK((call) == DST_OK)

/* src/encoding.rs:4-41 for the letters these fixtures use */
static uint8_t enc(char c)
{
    switch (c) {
    case 'A': return 136;
    case 'G': return 72;
    case 'C': return 40;
    case 'T': return 24;
    default: return 240;
    }
}

static void encode(const char *s, uint8_t *out)
{
    for (size_t i = 0; s[i]; ++i)
        out[i] = enc(s[i]);
}

int main(void)
{
    /* measures.rs:202-208 */
    const char *TARGET = "ATGATGATGATGCCC", *QUERY = "ATTATTATGATGCCC";
    uint8_t codes[2][15];
    encode(TARGET, codes[0]);
    encode(QUERY, codes[1]);
    dst_ctx *ctx = NULL;
    OK(dst_create(0, &ctx));
    if (!ctx) {
        fprintf(stderr, "%s\n", dst_last_error(NULL));
        return 1;
    }
    OK(dst_upload(ctx, 0, &codes[0][0], 2, 15, 15, NULL));

    /* test_snp / test_snp_consensus: FloatInt::Int(2) (measures.rs:219-238) */
    int64_t d_int = -1;
    OK(dst_run_square_host(ctx, DST_N_HIGH, 0, 2, DST_OUT_DISTANCE, &d_int, sizeof d_int));
    CHECK(d_int == 2);
    d_int = -1;
    OK(dst_run_square_host(ctx, DST_N, 0, 2, DST_OUT_DISTANCE, &d_int, sizeof d_int));
    CHECK(d_int == 2);

    /* test_raw: 2.0 / 15.0 (measures.rs:240-245): device value and exact host finalisation */
    double d = 0, host = 0;
    uint32_t t2[2], t3[3], t4[4], counts[8];
    OK(dst_run_square_host(ctx, DST_RAW, 0, 2, DST_OUT_DISTANCE, &d, sizeof d));
    CHECK(d == 2.0 / 15.0);
    OK(dst_run_square_host(ctx, DST_RAW, 0, 2, DST_OUT_TALLY, t2, sizeof t2));
    CHECK(t2[0] == 2 && t2[1] == 15);
    OK(dst_finalize(DST_RAW, t2, NULL, NULL, &host, NULL));
    CHECK(host == 2.0 / 15.0);

    /* test_jc69 (measures.rs:247-255) */
    const double want_jc = -0.75 * log(1.0 - (4.0 / 3.0) * (2.0 / 15.0));
    OK(dst_run_square_host(ctx, DST_JC69, 0, 2, DST_OUT_DISTANCE, &d, sizeof d));
    CHECK(fabs(d - want_jc) <= 1e-12);
    OK(dst_finalize(DST_JC69, t2, NULL, NULL, &host, NULL));
    CHECK(host == want_jc);

    /* test_k80 (measures.rs:257-269): P = 0, Q = 2/15 */
    const double P = 0.0 / 15.0, Q = 2.0 / 15.0;
    const double want_k80 = -0.5 * log((1.0 - 2.0 * P - Q) * sqrt(1.0 - 2.0 * Q));
    OK(dst_run_square_host(ctx, DST_K80, 0, 2, DST_OUT_DISTANCE, &d, sizeof d));
    CHECK(fabs(d - want_k80) <= 1e-12);
    OK(dst_run_square_host(ctx, DST_K80, 0, 2, DST_OUT_TALLY, t3, sizeof t3));
    CHECK(t3[0] == 15 && t3[1] == 0 && t3[2] == 2);
    OK(dst_finalize(DST_K80, t3, NULL, NULL, &host, NULL));
    CHECK(host == want_k80);

    /* test_tn93 (measures.rs:271-308) */
    const double g_A = 8.0 / 30.0, g_T = 10.0 / 30.0, g_C = 6.0 / 30.0, g_G = 6.0 / 30.0;
    const double g_R = (8.0 + 6.0) / 30.0, g_Y = (7.0 + 9.0) / 30.0;
    const double k1 = 2.0 * g_A * g_G / g_R, k2 = 2.0 * g_T * g_C / g_Y;
    const double k3 = 2.0 * (g_R * g_Y - g_A * g_G * g_Y / g_R - g_T * g_C * g_R / g_Y);
    const double P1 = 0.0 / 15.0, P2 = 0.0 / 15.0, Q3 = (2.0 - (0.0 + 0.0)) / 15.0;
    const double w1 = 1.0 - P1 / k1 - Q3 / (2.0 * g_R), w2 = 1.0 - P2 / k2 - Q3 / (2.0 * g_Y);
    const double w3 = 1.0 - Q3 / (2.0 * g_R * g_Y);
    const double want_tn = -k1 * log(w1) - k2 * log(w2) - k3 * log(w3);
    OK(dst_run_square_host(ctx, DST_TN93, 0, 2, DST_OUT_DISTANCE, &d, sizeof d));
    CHECK(fabs(d - want_tn) <= 1e-12);
    OK(dst_get_base_counts(ctx, 0, counts));
    CHECK(counts[0] == 4 && counts[1] == 4 && counts[2] == 4 && counts[3] == 3); /* fastaio.rs:363-366 {A,T,G,C} */
    OK(dst_run_square_host(ctx, DST_TN93, 0, 2, DST_OUT_TALLY, t4, sizeof t4));
    CHECK(t4[0] == 15 && t4[1] == 2 && t4[2] == 0 && t4[3] == 0);
    OK(dst_finalize(DST_TN93, t4, counts, counts + 4, &host, NULL));
    CHECK(host == want_tn);

    /* test_integration_1..3 (lib.rs:906-1154): seq1/seq2 vs seqA, n and n_high, all three modes */
    uint8_t f1[2][6], f2[1][6];
    encode("ATGATG", f1[0]);
    encode("ATGATC", f1[1]);
    encode("ATGATG", f2[0]);
    OK(dst_upload(ctx, 0, &f1[0][0], 2, 6, 6, NULL));
    OK(dst_run_square_host(ctx, DST_N, 0, 2, DST_OUT_DISTANCE, &d_int, sizeof d_int));
    CHECK(d_int == 1); /* seq1 seq2 1 */
    OK(dst_upload(ctx, 1, &f2[0][0], 1, 6, 6, NULL));
    int64_t r2[2] = {-1, -1};
    OK(dst_run_rect_host(ctx, DST_N_HIGH, 0, 1, 0, 2, DST_OUT_DISTANCE, r2, sizeof r2)); /* two files */
    CHECK(r2[0] == 0 && r2[1] == 1);
    r2[0] = r2[1] = -1;
    OK(dst_run_rect_host(ctx, DST_N_HIGH, 1, 0, 0, 1, DST_OUT_DISTANCE, r2, sizeof r2)); /* stream order */
    CHECK(r2[0] == 0 && r2[1] == 1);
    char text[64];
    CHECK(dst_format_distance(DST_N_HIGH, 0.0, r2[1], text, sizeof text) == 1 && text[0] == '1');
    /* the same TSV lines written by the GPU (gather_write, lib.rs:612-644): "seq1\tseq2\t1\n" (one file) and
     * "seq1\tseqA\t0\nseq2\tseqA\t1\n" (two files), as test_integration_1 / _3 expect them (lib.rs:906-1154) */
    {
        const char ids1[] = "seq1seq2", ids2[] = "seqA";
        const uint64_t off1[3] = {0, 4, 8}, off2[2] = {0, 4};
        char tsv[128];
        size_t len = 0;
        OK(dst_set_ids(ctx, 0, ids1, off1, 2));
        OK(dst_set_ids(ctx, 1, ids2, off2, 1));
        OK(dst_text_square(ctx, DST_N, 0, 2, tsv, sizeof tsv, &len));
        CHECK(len == 12 && memcmp(tsv, "seq1\tseq2\t1\n", 12) == 0);
        OK(dst_text_rect(ctx, DST_N_HIGH, 0, 1, 0, 2, 0, tsv, sizeof tsv, &len));
        CHECK(len == 24 && memcmp(tsv, "seq1\tseqA\t0\nseq2\tseqA\t1\n", 24) == 0);
        OK(dst_text_square(ctx, DST_RAW, 0, 2, tsv, sizeof tsv, &len));   /* 1 difference / 6 sites, {:.12} */
        CHECK(len == 25 && memcmp(tsv, "seq1\tseq2\t0.166666666667\n", 25) == 0);
        CHECK(dst_text_square(ctx, DST_RAW, 0, 2, tsv, 10, &len) == DST_ERR_CAPACITY);
    }

    /* error behaviour: invalid code byte, capacity, bad measure */
    uint8_t bad[2][6];
    memcpy(bad, f1, sizeof bad);
    bad[1][4] = 0; /* what encode() would have rejected: fastaio.rs:111-113 */
    CHECK(dst_upload(ctx, 0, &bad[0][0], 2, 6, 6, NULL) == DST_ERR_INVALID_CODE);
    CHECK(strstr(dst_last_error(ctx), "record 1 at site 4") != NULL);
    OK(dst_upload(ctx, 0, &f1[0][0], 2, 6, 6, NULL));
    CHECK(dst_run_square_host(ctx, DST_RAW, 0, 2, DST_OUT_DISTANCE, &d, 4) == DST_ERR_CAPACITY);
    CHECK(dst_run_square_host(ctx, 17, 0, 2, DST_OUT_DISTANCE, &d, sizeof d) == DST_ERR_ARG);

    /* ---- the per-alignment precompute of -m n on the device --------------------------------------------- */
    /* test_consensus (fastaio.rs:424-454): {FASTA, OTHER} -> FASTA's codes (1-1 ties at sites 2, 5 resolve to G);
     * {OTHER, OTHER} -> OTHER's codes */
    uint8_t two[2][15], cons[15];
    encode(TARGET, two[0]);
    encode(QUERY, two[1]);
    OK(dst_upload(ctx, 0, &two[0][0], 2, 15, 15, NULL));
    OK(dst_consensus(ctx, 0, cons, sizeof cons));
    CHECK(memcmp(cons, two[0], 15) == 0);
    memcpy(two[0], two[1], 15);
    OK(dst_upload(ctx, 0, &two[0][0], 2, 15, 15, NULL));
    OK(dst_consensus(ctx, 0, cons, sizeof cons));
    CHECK(memcmp(cons, two[1], 15) == 0);
    /* test_get_differences (fastaio.rs:369-377): FASTA vs OTHER -> [2, 5] */
    encode(TARGET, two[0]);
    OK(dst_upload(ctx, 0, &two[0][0], 1, 15, 15, NULL));
    uint64_t offs[2], total = 0;
    uint32_t sites[15];
    OK(dst_differences(ctx, 0, two[1], 15, offs, sites, 15, &total));
    CHECK(total == 2 && offs[0] == 0 && offs[1] == 2 && sites[0] == 2 && sites[1] == 5);

    /* ---- both kernel paths give the reference's integers (test_snp, test_snp_consensus) ---------------- */
    encode(TARGET, two[0]);
    encode(QUERY, two[1]);
    OK(dst_upload(ctx, 0, &two[0][0], 2, 15, 15, NULL));
    for (int path = DST_PATH_AUTO; path <= DST_PATH_CONSENSUS; ++path) {
        OK(dst_set_path(ctx, path));
        d_int = -1;
        OK(dst_run_square_host(ctx, DST_N, 0, 2, DST_OUT_DISTANCE, &d_int, sizeof d_int));
        CHECK(d_int == 2);
        OK(dst_run_square_host(ctx, DST_TN93, 0, 2, DST_OUT_TALLY, t4, sizeof t4));
        CHECK(t4[0] == 15 && t4[1] == 2 && t4[2] == 0 && t4[3] == 0);
        if (path != DST_PATH_AUTO)
            CHECK(dst_last_path(ctx) == path);
    }
    OK(dst_set_path(ctx, DST_PATH_AUTO));

    /* ---- stream mode through the overlapped pipeline: test_integration_2 (lib.rs:1000-1060) ------------- */
    OK(dst_upload(ctx, 0, &f1[0][0], 2, 6, 6, NULL));
    dst_stream *st = NULL;
    OK(dst_stream_open(ctx, DST_N_HIGH, DST_OUT_DISTANCE, 4, 2, &st));
    uint8_t *buf = NULL;
    size_t pitch = 0, n_rec = 0;
    const void *res = NULL;
    OK(dst_stream_acquire(st, &buf, &pitch, NULL));
    CHECK(pitch >= 6);
    encode("ATGATG", buf); /* seqA */
    OK(dst_stream_submit(st, 1, 0));
    CHECK(dst_stream_in_flight(st) == 1);
    OK(dst_stream_collect(st, &n_rec, &res));
    CHECK(n_rec == 1 && ((const int64_t *)res)[0] == 0 && ((const int64_t *)res)[1] == 1); /* seq1 seqA 0 / seq2 seqA 1 */
    OK(dst_stream_close(st));

    /* ---- the gather of result slabs: a one-rank communicator (RCCL itself needs one GPU per rank) ------- */
    {
        uint8_t id[DST_COMM_ID_BYTES];
        dst_comm *comm = NULL;
        OK(dst_comm_unique_id(id, sizeof id));
        OK(dst_comm_create(ctx, id, 0, 1, &comm));
        int rank = -1, world = -1;
        OK(dst_comm_info(comm, &rank, &world));
        CHECK(rank == 0 && world == 1);
        void *d_local = NULL, *d_full = NULL;
        /* device buffers through the run API: rows [0,2) of the 2-record set = 1 pair */
        CHECK(dst_host_alloc(16, &d_local) == DST_OK); /* pinned host memory is device-accessible: good enough as a slab */
        CHECK(dst_host_alloc(16, &d_full) == DST_OK);
        ((int64_t *)d_local)[0] = 42;
        ((int64_t *)d_full)[0] = -1;
        const uint64_t goff[1] = {0}, gsize[1] = {8};
        OK(dst_gather_slabs(comm, d_local, d_full, goff, gsize, 0, NULL));
        CHECK(((int64_t *)d_full)[0] == 42);
        CHECK(dst_gather_slabs(comm, d_local, d_full, goff, gsize, 3, NULL) == DST_ERR_ARG);
        dst_host_free(d_local);
        dst_host_free(d_full);
        OK(dst_comm_destroy(comm));
    }
    OK(dst_destroy(ctx));
    if (!failures)
        puts("abi_reference_tests: all checks passed");
    return failures;
}
