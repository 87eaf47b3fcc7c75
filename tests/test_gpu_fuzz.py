"""GPU (-m gpu): seeded randomised sweep over shapes, measures, modes, row ranges, kernel paths (dense
bit-planes / consensus-delta), tile variants and split-L factors — integer tallies bit-exact against the oracle, device distances within 1e-12,
host-finalised distances bit-identical."""
import math

import numpy as np
import pytest

import distance_amd as da
import oracle
from helpers import CODES, random_alignment, uniform_codes

pytestmark = pytest.mark.gpu
MEASURES = ("n", "n_high", "raw", "jc69", "k80", "tn93")


def _alignment(rng, n, L, fused=False):
    kind = rng.integers(0, 7 if fused else 4)
    seed = int(rng.integers(0, 2**31))
    if kind >= 4:
        # low diversity around the gate of the fused preparation (8 % of the sites deviating): lists counted and slotted
        # by the pack, chunks with more than 7 differences (back to the planes), lean sets, a few hot columns
        a = random_alignment(n, L, seed, p_ambig=float(rng.choice([0.0, 0.002])), p_gap=float(rng.choice([0.0, 0.01])),
                             divergence=float(rng.choice([0.002, 0.02, 0.05, 0.07])))
        if L > 40 and n > 8 and kind == 6:
            cols = rng.choice(L, size=max(1, L // 60), replace=False)     # clade-like columns: hot sites
            rows = rng.random(n) < 0.4
            a[np.ix_(rows, cols)] = CODES[int(rng.integers(0, 4))]
        return a
    if kind == 0:
        return uniform_codes(n, L, seed)
    if kind == 1:
        return random_alignment(n, L, seed, p_ambig=0.0, p_gap=0.0, divergence=float(rng.random()) * 0.5)
    if kind == 2:
        a = random_alignment(n, L, seed, p_ambig=0.05, p_gap=0.3, divergence=0.02)
        if L:
            a[:, : L // 3] = 244          # long leading gap run
        return a
    return random_alignment(n, L, seed)


def _close(got, want):
    if math.isnan(want):
        return math.isnan(got)
    if math.isinf(want):
        return got == want
    return abs(got - want) <= 1e-12 and abs(got - want) <= 1e-12 * max(abs(want), 1e-300) + 1e-300 or abs(got - want) <= 1e-12


@pytest.mark.parametrize("fused", [False, True])
def test_randomised_parity_sweep(fused):
    """fused: every upload goes through the consensus path's fused preparation (dst_set_prep_threshold(0)) — the
    reference sampled from the bytes, list lengths and slots from the pack, lean sets — which the default threshold
    only gives to jobs of 2e10 site comparisons and more."""
    rng = np.random.default_rng(20261004 + (77 if fused else 0))
    eng = da.Engine(0)
    if fused:
        eng.set_prep_threshold(0.0)
    lib = da.load()
    cases = 0
    try:
        for _ in range(60):
            n = int(rng.choice([1, 2, 3, 17, 64, 65, 200, 513, 700]))
            L = int(rng.choice([0, 1, 31, 32, 33, 127, 128, 129, 500, 2049, 4100]))
            a = _alignment(rng, n, L, fused)
            eng.upload(0, a)
            two = bool(rng.integers(0, 2))
            if two:
                nb = int(rng.choice([1, 5, 130]))
                b = _alignment(rng, nb, L, fused)
                eng.upload(1, b)
            ca = oracle.count_bases_matrix(a)
            for m in rng.choice(MEASURES, size=3, replace=False):
                m = str(m)
                eng.set_variant(int(rng.integers(0, lib.dst_variant_count(da.MEASURES[m]))))
                eng.set_ksplit(int(rng.choice([0, 0, 1, 3, 16])))
                eng.set_path(str(rng.choice(["dense", "consensus", "hybrid", "hybrid", "auto"])))
                om = "n_high" if m == "n" else m
                if not two:
                    rb = int(rng.integers(0, n))
                    re = int(rng.integers(rb, n + 1))
                    lo, hi = da.square_row_start(n, min(rb, n)), da.square_row_start(n, min(re, n))
                    want = oracle.all_pairs_square(m, a)[lo:hi]
                    ij = oracle.pairs_square(n)[lo:hi]
                    got = eng.run_square(m, rb, re)
                    tl = eng.run_square(m, rb, re, tallies=True)
                    pairs = [(int(i), int(j), a[int(i)], a[int(j)], ca[int(i)], ca[int(j)]) for i, j in ij]
                else:
                    cb = oracle.count_bases_matrix(b)
                    stream_order = bool(rng.integers(0, 2))
                    w = oracle.all_pairs_rect(m, a, b)
                    if stream_order:
                        got = eng.run_rect(m, row_slot=1, col_slot=0).ravel()
                        tl = eng.run_rect(m, row_slot=1, col_slot=0, tallies=True).reshape(-1, da.tally_width(m))
                        want = w.T.ravel()
                        pairs = [(i, j, a[i], b[j], ca[i], cb[j]) for j in range(nb) for i in range(n)]
                    else:
                        got = eng.run_rect(m).ravel()
                        tl = eng.run_rect(m, tallies=True).reshape(-1, da.tally_width(m))
                        want = w.ravel()
                        pairs = [(i, j, a[i], b[j], ca[i], cb[j]) for i in range(n) for j in range(nb)]
                assert len(got) == len(want) == len(tl) == len(pairs), (n, L, m)
                if m in da.INT_MEASURES:
                    assert np.array_equal(got, want.astype(np.int64)), (n, L, m)
                else:
                    bad = [k for k in range(len(want)) if not _close(float(got[k]), float(want[k]))]
                    assert not bad, (n, L, m, bad[:3])
                # tallies bit-exact (sample when large), host finalisation bit-identical
                idx = range(len(pairs)) if len(pairs) <= 400 else rng.choice(len(pairs), 400, replace=False)
                for k in idx:
                    _, _, q, t, qc, tc = pairs[int(k)]
                    assert list(tl[int(k)]) == list(oracle.tallies(om, q, t)), (n, L, m, int(k))
                    host = da.finalize(m, tl[int(k)], qc, tc)
                    ref = oracle.pair_distance(om, q, t, q_counts=qc, t_counts=tc)
                    assert host == ref or (math.isnan(host) and math.isnan(ref)), (n, L, m, int(k))
                cases += 1
    finally:
        eng.close()
    assert cases == 180
