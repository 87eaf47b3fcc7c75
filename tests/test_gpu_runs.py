"""GPU (-m gpu): records with long runs of N (failed amplicons, partial genomes) on the consensus path.

N contributes nothing to any tally (src/measures.rs:17, 59-66, 89-107, 160-175), but every site of a run is a difference
from the reference sequence, and two such records met at every site where their runs overlap.  The engine leaves the
128-site chunks of N of "run records" out of their lists and corrects every pair with such a record exactly
(dst_internal.h: RunIndex).  Here: every measure, tallies and distances, bit for bit against the dense kernels (which
know nothing of lists) and sampled against the oracle — runs of every length and alignment, records below the
threshold, an all-N record, runs inside hot columns (the hybrid path), two files after a square run, the TSV text."""
import numpy as np
import pytest

import distance_amd as da
import oracle
from tools import synth

pytestmark = pytest.mark.gpu
ALL = ("n", "n_high", "raw", "jc69", "k80", "tn93")


@pytest.fixture(scope="module")
def eng():
    e = da.Engine(0)
    yield e
    e.close()


def with_runs(n, L, seed, share, frac, extra=True):
    rng = np.random.default_rng(seed)
    codes = synth.alignment(synth.SEED ^ seed, n, L)
    synth.apply_nruns(codes, synth.nrun_plan(seed, n, L, share, frac))
    if extra:
        codes[5, :] = 240                                    # a record of nothing but N
        codes[6, 128 * 3:128 * 6] = 244                      # exactly three run chunks of gaps: below the threshold
        codes[7, 100:100 + 128 * 9 + 17] = 242               # unaligned: partial chunks at both ends stay in the list
        codes[8, L - 700:] = 240                             # to the end of the alignment (the last chunk is partial)
        codes[9, :650] = 240
        for r in rng.choice(n, 12, replace=False):           # short runs everywhere: never a whole chunk
            a = int(rng.integers(0, L - 100))
            codes[r, a:a + int(rng.integers(5, 100))] = 240
    return codes


def dense_reference(codes, measures, rows):
    with da.Engine(0) as ref:
        ref.set_path("dense")
        ref.upload(0, codes)
        out = {(m, r): ref.run_square(m, r, r + 1) for m in measures for r in rows}
        tal = {(m, r): ref.run_square(m, r, r + 1, tallies=True) for m in ("raw", "k80", "tn93") for r in rows}
    return out, tal


@pytest.mark.parametrize("share,frac", [(0.05, 0.5), (0.2, 0.1), (0.3, 0.3)])
def test_run_records_give_the_dense_bits(eng, share, frac):
    n, L = 2_500, 6_000
    codes = with_runs(n, L, 41, share, frac)
    plan_rows = sorted({r for r, _, _ in synth.nrun_plan(41, n, L, share, frac)})
    rows = sorted({0, 5, 6, 7, 8, 9, n - 2, plan_rows[0], plan_rows[len(plan_rows) // 2], plan_rows[-1]} - {n - 1})
    want, want_t = dense_reference(codes, ALL, rows)
    eng.set_prep_threshold(0)
    for path in ("consensus", "hybrid", "auto"):
        eng.set_path("auto")
        eng.upload(0, codes)
        eng.set_path(path)
        for (m, r), w in want.items():
            assert np.array_equal(eng.run_square(m, r, r + 1), w, equal_nan=True), (path, m, r)
        for (m, r), w in want_t.items():
            assert np.array_equal(eng.run_square(m, r, r + 1, tallies=True), w), (path, m, r)
        n_run, removed = eng.run_records()
        if path == "consensus":
            assert eng.last_path() == "consensus"
            # the stripping is what ran: every record with four or more whole 128-site chunks of N is a run record
            nch = (L + 127) // 128                      # (sites past the alignment's end count as N, like on the device)
            padded = np.full((n, nch * 128), 240, np.uint8)
            padded[:, :L] = codes
            whole = ((padded >> 4) == 15).reshape(n, nch, 128).all(axis=2).sum(axis=1)
            assert n_run == int((whole >= 4).sum()) > 0 and removed > 0, (n_run, removed)
    # the dense reference itself against the oracle: run record x run record, run x plain, plain x run
    rng = np.random.default_rng(1)
    for r in rows[:6]:
        for j in sorted({int(x) for x in rng.integers(r + 1, n, 4)} | {plan_rows[-1]} - set(range(r + 1))):
            assert list(want_t[("tn93", r)][j - r - 1]) == [int(x) for x in oracle.tallies("tn93", codes[r], codes[j])], (r, j)
    eng.set_prep_threshold(2e10)
    eng.set_path("auto")


def test_whole_triangle_and_the_plain_preparation(eng):
    """every pair of a smaller set, fused preparation (strips) and lists built at the first run (does not): same bits"""
    n, L = 700, 3_000
    codes = with_runs(n, L, 43, 0.1, 0.4)
    with da.Engine(0) as ref:
        ref.set_path("dense")
        ref.upload(0, codes)
        want = {m: ref.run_square(m) for m in ALL}
    for threshold, strips in ((0.0, True), (1e30, False)):
        eng.set_prep_threshold(threshold)
        eng.set_path("auto")
        eng.upload(0, codes)
        eng.set_path("consensus")
        for m in ALL:
            assert np.array_equal(eng.run_square(m), want[m], equal_nan=True), (threshold, m)
        assert (eng.run_records()[0] > 0) == strips
    eng.set_prep_threshold(2e10)
    eng.set_path("auto")
    d = oracle.all_pairs_square("tn93", codes, threads=8)
    assert np.allclose(want["tn93"], d, rtol=0, atol=1e-12, equal_nan=True)


def test_two_files_after_a_square_run_and_the_text(eng):
    """the stripped lists serve the square job; a second file against the same set rebuilds them whole"""
    n, L = 900, 4_000
    a = with_runs(n, L, 47, 0.08, 0.5)
    b = with_runs(300, L, 48, 0.1, 0.3, extra=False)
    ids = ["r%d" % k for k in range(n)]
    eng.set_prep_threshold(0)
    eng.set_path("auto")
    eng.upload(0, a)
    eng.set_path("consensus")
    sq = eng.run_square("k80", 0, 40)
    assert eng.run_records()[0] > 0
    eng.set_ids(0, ids)
    text = eng.text_square("tn93", 10, 14, capacity=1 << 24)
    tl = eng.run_square("tn93", 10, 14, tallies=True)
    host = oracle.finalize_square("tn93", tl, n, oracle.count_bases_matrix(a), 10, 14)
    assert text == oracle.tsv_square("tn93", host, ids, 10, 14)
    eng.upload(1, b)
    rect = eng.run_rect("k80", 1, 0)                          # rows: the second file, columns: the stripped set
    assert eng.run_records()[0] == 0                          # ... whose lists were rebuilt whole
    with da.Engine(0) as ref:
        ref.set_path("dense")
        ref.upload(0, a)
        ref.upload(1, b)
        assert np.array_equal(sq, ref.run_square("k80", 0, 40), equal_nan=True)
        assert np.array_equal(rect, ref.run_rect("k80", 1, 0), equal_nan=True)
    assert np.array_equal(eng.run_square("k80", 0, 40), sq, equal_nan=True)   # and the square job still answers the same
    eng.set_prep_threshold(2e10)
    eng.set_path("auto")


def test_hot_columns_and_run_records_together(eng):
    """clade-like columns (the hybrid path hands them to the dense kernels, the lists leave them out) AND records with runs of
    N that cross them: F over a run chunk then counts the COLD known sites only, the dense part sees the N as they are"""
    n, L = 1_800, 5_000
    codes = with_runs(n, L, 51, 0.08, 0.4)
    codes[: n // 3, ::40] = np.where(codes[: n // 3, ::40] >= 240, codes[: n // 3, ::40], 72)   # a third of the records share G there
    rows = [0, 5, 7, n // 3 - 1, n // 3, n - 2]
    want, want_t = dense_reference(codes, ALL, rows)
    eng.set_prep_threshold(0)
    for path in ("hybrid", "consensus", "auto"):
        eng.set_path("auto")
        eng.upload(0, codes)
        eng.set_path(path)
        for (m, r), w in want.items():
            assert np.array_equal(eng.run_square(m, r, r + 1), w, equal_nan=True), (path, m, r)
        for (m, r), w in want_t.items():
            assert np.array_equal(eng.run_square(m, r, r + 1, tallies=True), w), (path, m, r)
        if path == "hybrid":
            assert eng.last_path() == "hybrid" and eng.run_records()[0] > 0
    eng.set_prep_threshold(2e10)
    eng.set_path("auto")


def test_wide_alignment_one_word_per_tally(eng):
    """L >= 65,536: 32-bit accumulators (up to four words per pair: the tables' matrix product per word), masks of 18 words
    (three rounds of eight k-steps)"""
    n, L = 420, 70_000
    codes = with_runs(n, L, 53, 0.15, 0.5, extra=False)
    rows = [0, 1, n // 2, n - 2]
    want, want_t = dense_reference(codes, ALL, rows)
    eng.set_prep_threshold(0)
    eng.set_path("auto")
    eng.upload(0, codes)
    eng.set_path("consensus")
    for (m, r), w in want.items():
        assert np.array_equal(eng.run_square(m, r, r + 1), w, equal_nan=True), (m, r)
    for (m, r), w in want_t.items():
        assert np.array_equal(eng.run_square(m, r, r + 1, tallies=True), w), (m, r)
    assert eng.run_records()[0] > 0
    eng.set_prep_threshold(2e10)
    eng.set_path("auto")


def test_a_fifth_of_fifty_thousand_records_with_runs_keep_their_run_records():
    """50,000 records of which a fifth carry runs of N: 10,000 run records, whose correction tables (two x n_run x n words)
    the r03 budget (24 GB) holds — at 6 GB they fell back to whole lists and took 35 ms where 20,000 such records took 1.7.
    Rows of run records and of others against the dense kernels' bits and the oracle."""
    n, L = 50_000, 4_096
    codes = synth.alignment(synth.SEED ^ 43, n, L)
    plan = synth.nrun_plan(43, n, L, 0.2, 0.25)
    synth.apply_nruns(codes, plan)
    run_rows = sorted({r for r, _, _ in plan})
    rows = sorted({0, run_rows[0], run_rows[len(run_rows) // 2], run_rows[-1], n // 3, n - 2} - {n - 1})
    want, _ = dense_reference(codes, ("raw", "tn93"), rows)
    with da.Engine(0) as eng:
        eng.set_prep_threshold(0)
        eng.set_path("consensus")
        eng.upload(0, codes)
        for m in ("raw", "tn93"):
            for r in rows:
                assert np.array_equal(eng.run_square(m, r, r + 1), want[(m, r)], equal_nan=True), (m, r)
        n_run, removed = eng.run_records(0)
        assert n_run > 5_000 and removed > 0, (n_run, removed)    # (more than the 3,750 a 6 GB budget allowed)
        rng = np.random.default_rng(3)
        for i in (run_rows[0], n // 3):
            for j in rng.integers(i + 1, n, 6):
                j = int(j)
                got = eng.run_square("raw", i, i + 1)[j - i - 1]
                assert got == oracle.pair_distance("raw", codes[i], codes[j]) or (np.isnan(got) and np.isnan(oracle.pair_distance("raw", codes[i], codes[j])))
