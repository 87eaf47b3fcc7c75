"""GPU (-m gpu): a set prepared for the consensus path does not have its bit-planes stored by the upload — only the
chunks that do not fit their 16-byte slot of differences (dst_planes_stored).  Everything that reads planes
(the dense and hybrid paths, base counts for tn93, dst_consensus, dst_differences, a run against another set) must see
exactly the planes a full pack would have written: here each of them after a deferring upload, bit for bit against an
engine that packs every plane (path "dense" at upload) and sampled against the oracle (src/measures.rs, src/fastaio.rs:53-66,
289-336).  The inputs mix everything a chunk can be: no difference, a few, more than a slot holds (a dense patch of
substitutions), chunks of N (in run records and in records below the run threshold), ambiguity codes, gaps, a partial last
chunk."""
import numpy as np
import pytest

import distance_amd as da
import oracle
from tools import synth

pytestmark = pytest.mark.gpu
ALL = ("n", "n_high", "raw", "jc69", "k80", "tn93")


def mixed_alignment(n, L, seed):
    rng = np.random.default_rng(seed)
    codes = synth.alignment(synth.SEED ^ seed, n, L)
    synth.apply_nruns(codes, synth.nrun_plan(seed, n, L, 0.05, 0.4))
    bases = np.array([136, 72, 40, 24], np.uint8)
    for r in rng.choice(n, 40, replace=False):               # a patch denser than a slot: 20-60 substitutions in 128 sites
        a = int(rng.integers(0, L - 128))
        at = a + rng.choice(128, int(rng.integers(20, 60)), replace=False)
        codes[r, at] = bases[rng.integers(0, 4, len(at))]
    amb = np.array([192, 160, 144, 96, 80, 48, 224, 208, 176, 112, 240, 244, 242], np.uint8)
    for r in rng.choice(n, 60, replace=False):               # ambiguity codes, gaps and ? scattered
        at = rng.choice(L, 25, replace=False)
        codes[r, at] = amb[rng.integers(0, len(amb), len(at))]
    codes[3, :] = 240                                        # nothing but N
    codes[4, 128 * 2:128 * 5] = 244                          # three chunks of gaps: below the run-record threshold
    codes[5, L - 300:] = 240                                 # into the partial last chunk
    return codes


def dense_engine_results(codes, measures, rows, other=None):
    with da.Engine(0) as ref:
        ref.set_path("dense")
        ref.upload(0, codes)
        assert ref.planes_stored(0)
        out = {(m, r): ref.run_square(m, r, r + 1) for m in measures for r in rows}
        counts = ref.base_counts(0)
        cons = ref.consensus()
        rect = None
        if other is not None:
            ref.upload(1, other)
            rect = {m: ref.run_rect(m, 1, 0) for m in measures}
    return out, counts, cons, rect


@pytest.mark.parametrize("n,L", [(2_000, 5_000), (1_100, 3_333)])
def test_every_reader_of_planes_after_a_deferring_upload(n, L):
    codes = mixed_alignment(n, L, 77)
    other = codes[rng_rows(n, 40)].copy()
    rows = sorted({0, 3, 4, 5, n // 2, n - 2})
    want, want_counts, want_cons, want_rect = dense_engine_results(codes, ALL, rows, other)
    with da.Engine(0) as eng:
        eng.set_prep_threshold(0)
        eng.upload(0, codes)
        assert not eng.planes_stored(0), "the upload stored every plane: nothing here tests the deferred form"
        # the consensus path never asks for planes
        eng.set_path("consensus")
        for m in ALL:
            for r in rows:
                assert np.array_equal(eng.run_square(m, r, r + 1), want[(m, r)], equal_nan=True), (m, r)
        assert eng.last_path() == "consensus"
        # base counts of a deferred set come from the slots (tn93 ran above with them)
        assert not eng.planes_stored(0)
        assert np.array_equal(eng.base_counts(0), want_counts)
        assert np.array_equal(eng.base_counts(0)[:60], np.stack([oracle.count_bases(r) for r in codes[:60]]))
        # the dense kernels: the planes are written now
        eng.set_path("dense")
        for m in ALL:
            for r in rows:
                assert np.array_equal(eng.run_square(m, r, r + 1), want[(m, r)], equal_nan=True), (m, r)
        assert eng.planes_stored(0) and eng.last_path() == "dense"
        rng = np.random.default_rng(5)
        for _ in range(30):
            i = rows[int(rng.integers(0, len(rows)))]
            j = int(rng.integers(i + 1, n))
            assert int(eng.run_square("n_high", i, i + 1)[j - i - 1]) == oracle.pair_distance("n_high", codes[i], codes[j])


def rng_rows(n, k):
    return np.sort(np.random.default_rng(n).choice(n, k, replace=False))


def test_consensus_differences_and_two_files_after_a_deferring_upload():
    n, L = 1_500, 4_000
    codes = mixed_alignment(n, L, 78)
    other = codes[rng_rows(n, 50)].copy()
    other[:, 100:140] = 136
    _, _, want_cons, want_rect = dense_engine_results(codes, ALL, [0], other)
    assert np.array_equal(want_cons, oracle.consensus(codes))
    for first in ("consensus", "differences", "rect", "hybrid"):
        with da.Engine(0) as eng:
            eng.set_prep_threshold(0)
            eng.upload(0, codes)
            assert not eng.planes_stored(0)
            if first == "consensus":
                assert np.array_equal(eng.consensus(), want_cons)
            elif first == "differences":
                got = eng.differences(0, want_cons)
                for r in (0, 3, 4, 5, 700, n - 1):
                    assert np.array_equal(got[r].astype(np.uint64), oracle.get_differences(codes[r], want_cons)), r
            elif first == "rect":
                eng.upload(1, other)
                for m in ALL:
                    assert np.array_equal(eng.run_rect(m, 1, 0), want_rect[m], equal_nan=True), m
            else:
                eng.set_path("hybrid")
                a = eng.run_square("tn93", 0, 50)
                eng.set_path("dense")
                assert np.array_equal(a, eng.run_square("tn93", 0, 50), equal_nan=True)
            if first != "rect":   # (two files on the consensus path list the row set against the column set's reference: no planes)
                assert eng.planes_stored(0) or first == "hybrid"


def test_a_diverse_set_keeps_its_planes():
    from helpers import random_alignment
    codes = random_alignment(600, 2_000, 9, divergence=0.3)
    with da.Engine(0) as eng:
        eng.set_prep_threshold(0)
        eng.upload(0, codes)
        assert eng.planes_stored(0)
        i, j = 17, 333
        assert int(eng.run_square("n_high", i, i + 1)[j - i - 1]) == oracle.pair_distance("n_high", codes[i], codes[j])
