"""GPU (-m gpu): the HIP path at BASELINE.json's full sizes, on SURVEY §8(d)'s synthetic alignments
(tools/synth: xoshiro256**, the generator bench.py uses), sampled against the oracle — tallies bit-exact,
device distances within 1e-12 — on BOTH kernel paths:

  C1  100 x 10,000   -m n      complete, against the sparse walk      (test_gpu_consensus.py)
  C2  10,000 x 30,000 -m raw   every row range, pairs sampled over the whole triangle
  C3  50,000 x 30,000 -m raw and tn93, rows {0, 24999, 49998} and sampled pairs
  C4  1,000 x 5,000,000 loaded vs a 64-record streamed batch, -m n_high
  C5  200,000 x 1,000 -m jc69  (test_gpu_parity.py::test_two_hundred_thousand_records)
"""
import math

import numpy as np
import pytest

import distance_amd as da
import oracle
from tools import synth

pytestmark = pytest.mark.gpu
TOL = 1e-12


def close(got, want):
    return (math.isnan(want) and math.isnan(got)) or got == want or abs(got - want) <= TOL


@pytest.fixture(scope="module")
def eng():
    e = da.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def c3_codes():
    return synth.alignment(synth.SEED ^ 3, 50_000, 30_000)


def check_rows(eng, codes, measure, rows, rng, per_row=18, counts=None):
    n = len(codes)
    om = measure
    for i in rows:
        tl = eng.run_square(measure, i, i + 1, tallies=True)
        d = eng.run_square(measure, i, i + 1)
        assert len(d) == n - i - 1
        js = sorted({i + 1, n - 1} | {int(x) for x in rng.integers(i + 1, n, per_row)}) if i + 1 < n else []
        for j in js:
            assert list(tl[j - i - 1]) == [int(x) for x in oracle.tallies(om, codes[i], codes[j])], (measure, i, j)
            want = oracle.pair_distance(om, codes[i], codes[j])
            assert close(float(d[j - i - 1]), want), (measure, i, j, d[j - i - 1], want)


@pytest.mark.parametrize("path", ["dense", "consensus"])
def test_c3_50000_x_30000_raw_and_tn93(eng, c3_codes, path):
    codes = c3_codes
    eng.set_path(path)
    eng.upload(0, codes)
    rng = np.random.default_rng(33)
    for m in ("raw", "tn93"):
        check_rows(eng, codes, m, [0, 24_999, 49_998], rng)
        assert eng.last_path() == path
    # tn93's base counts at full size
    bc = eng.base_counts(0)
    for r in (0, 31_337, 49_999):
        assert list(bc[r]) == [int(x) for x in oracle.count_bases(codes[r])]
    eng.set_path("auto")
    if path == "consensus":
        # SARS-CoV-2-like diversity: with the lists in place, the default choice for a 10M-pair launch is the lists
        got = eng.run_square("raw", 0, 200)
        assert eng.last_path() == "consensus"
        eng.set_path("dense")
        assert np.array_equal(eng.run_square("raw", 0, 200), got, equal_nan=True)
        eng.set_path("auto")


@pytest.mark.parametrize("path", ["dense", "consensus", "hybrid"])
def test_c2_10000_x_30000_raw_whole_triangle(eng, path):
    n, L = 10_000, 30_000
    codes = synth.alignment(synth.SEED ^ 2, n, L)
    eng.set_path(path)
    eng.upload(0, codes)
    d = eng.run_square("raw")                        # all 49,995,000 pairs
    tl = eng.run_square("raw", tallies=True)
    # (the synthetic star phylogeny has no hot columns: "hybrid" runs as the consensus path there)
    assert d.shape == (n * (n - 1) // 2,) and eng.last_path() == ("consensus" if path == "hybrid" else path)
    with np.errstate(invalid="ignore", divide="ignore"):
        want = tl[:, 0] / tl[:, 1].astype(np.float64)
    assert np.array_equal(d, want, equal_nan=True)    # IEEE division: device == host on the same tallies
    rng = np.random.default_rng(22)
    for _ in range(300):
        i = int(rng.integers(0, n - 1))
        j = int(rng.integers(i + 1, n))
        at = da.square_row_start(n, i) + j - i - 1
        assert list(tl[at]) == [int(x) for x in oracle.tallies("raw", codes[i], codes[j])], (i, j)
        assert close(float(d[at]), oracle.pair_distance("raw", codes[i], codes[j]))
    eng.set_path("auto")


def test_c4_1000_x_5mbp_loaded_vs_streamed_batch(eng):
    """C4's shape: the loaded set resident once, one 64-record streamed batch, -m n_high, streamed-major."""
    L = 5_000_000
    seed = synth.SEED ^ 4
    root = synth.root(seed, L)
    loaded = synth.records(seed, root, 0, 1000)
    batch = synth.records(seed, root, 1000, 64)       # the next 64 records of the same alignment
    eng.set_path("auto")
    eng.upload(0, loaded)
    got = eng.run_stream_batch("n_high", batch)       # [streamed][loaded]
    assert got.shape == (64, 1000)
    rng = np.random.default_rng(44)
    for _ in range(40):
        s, i = int(rng.integers(0, 64)), int(rng.integers(0, 1000))
        assert int(got[s, i]) == oracle.pair_distance("n_high", loaded[i], batch[s]), (s, i)
    first = eng.last_path()
    other = "consensus" if first == "dense" else "dense"
    eng.set_path(other)
    again = eng.run_rect("n_high", row_slot=1, col_slot=0)
    assert eng.last_path() == other and np.array_equal(again, got)
    eng.set_path("auto")
