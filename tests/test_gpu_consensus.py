"""GPU (-m gpu): the consensus-delta path (dst_consensus.hip) and the per-alignment precompute of `-m n`
(consensus(), src/fastaio.rs:289-336; get_differences(), src/fastaio.rs:67-75) against the oracle.
The general parity suite (test_gpu_parity.py, test_gpu_fuzz.py) already runs every case on this path;
here are the shapes that are specific to it: panels of 2,048 columns, 4-row batches, lists longer than one slice,
32-bit tallies, shared gap runs, the path choice."""
import numpy as np
import pytest

import distance_amd as da
import oracle
from helpers import CODES, random_alignment, uniform_codes

pytestmark = pytest.mark.gpu
ALL = ("n", "n_high", "raw", "jc69", "k80", "tn93")


@pytest.fixture(scope="module")
def eng():
    e = da.Engine(0)
    yield e
    e.close()


def low_diversity(n, L, seed, subs=3e-3, p_n=2e-3, p_amb=3e-4, gap_rows=0.05):
    """SARS-CoV-2-like: one root, sparse substitutions, sparse N / IUPAC codes, shared terminal gap runs."""
    rng = np.random.default_rng(seed)
    root = rng.choice(np.array([136, 72, 40, 24], np.uint8), size=L)
    codes = np.tile(root, (n, 1))
    m = rng.random((n, L)) < subs
    codes[m] = rng.choice(np.array([136, 72, 40, 24], np.uint8), size=int(m.sum()))
    m = rng.random((n, L)) < p_n
    codes[m] = rng.choice(CODES[14:], size=int(m.sum()))
    m = rng.random((n, L)) < p_amb
    codes[m] = rng.choice(CODES[4:14], size=int(m.sum()))
    for r in np.nonzero(rng.random(n) < gap_rows)[0]:
        codes[r, : rng.integers(0, min(L, 60) + 1)] = 244
        codes[r, L - rng.integers(0, min(L, 60) + 1):] = 244
    return np.ascontiguousarray(codes)


def sample_check(eng, codes, measure, pairs, counts=None):
    """tallies bit-exact and distances within 1e-12 on sampled (i, j) of the square run"""
    n = len(codes)
    om = "n_high" if measure == "n" else measure
    by_row = {}
    for i, j in pairs:
        by_row.setdefault(int(i), []).append(int(j))
    for i, js in by_row.items():
        tl = eng.run_square(measure, i, i + 1, tallies=True)
        d = eng.run_square(measure, i, i + 1)
        for j in js:
            assert list(tl[j - i - 1]) == [int(x) for x in oracle.tallies(om, codes[i], codes[j])], (measure, i, j)
            want = oracle.pair_distance(om, codes[i], codes[j])
            got = d[j - i - 1]
            assert (np.isnan(want) and np.isnan(got)) or got == want or abs(got - want) <= 1e-12, (measure, i, j)


def test_paths_agree_bit_for_bit_and_report_themselves(eng):
    codes = low_diversity(700, 3000, 1)
    eng.upload(0, codes)
    for m in ALL:
        eng.set_path("dense")
        td, dd = eng.run_square(m, tallies=True), eng.run_square(m)
        assert eng.last_path() == "dense"
        eng.set_path("consensus")
        tc, dc = eng.run_square(m, tallies=True), eng.run_square(m)
        assert eng.last_path() == "consensus"
        assert np.array_equal(td, tc), m
        assert np.array_equal(dd, dc, equal_nan=True), m       # same device finalisation on the same tallies
        t16 = eng.run_square(m, tallies16=True)
        assert np.array_equal(t16.astype(np.uint32), tc), m
    eng.set_path("auto")


def test_column_panels_and_row_tiles(eng):
    """Many panels of 2,048 column records; rows that start inside a panel; odd and partial last panel."""
    n, L = 17001, 400
    codes = low_diversity(n, L, 2, subs=5e-3)
    eng.set_path("consensus")
    eng.upload(0, codes)
    rng = np.random.default_rng(4)
    rows = [0, 1, 2, 3, 4, 5, 31, 32, 33, 2046, 2047, 2048, 2049, 4095, 4096, 8191, 8192, 8193, 16383, 16384, 16385, n - 3, n - 2] + [int(x) for x in rng.integers(0, n - 1, 12)]
    pairs = []
    for i in rows:
        cols = {i + 1, min(n - 1, i + 2), n - 1, n - 2, min(n - 1, max(i + 1, 2047)), min(n - 1, max(i + 1, 2048)), min(n - 1, max(i + 1, 8192)),
                min(n - 1, max(i + 1, 16384))} | {int(x) for x in rng.integers(i + 1, n, 6)}
        pairs += [(i, j) for j in cols if j > i]
    for m in ("n_high", "raw", "tn93", "k80"):
        sample_check(eng, codes, m, pairs)
    assert eng.last_path() == "consensus"
    # the whole triangle agrees with the dense path (tallies, 144M pairs would be too many: compare a row band)
    band = (8100, 8300)
    tc = eng.run_square("raw", *band, tallies=True)
    eng.set_path("dense")
    assert np.array_equal(eng.run_square("raw", *band, tallies=True), tc)
    eng.set_path("auto")


def test_long_lists_and_dense_columns(eng):
    """Rows whose list is longer than one 256-entry slice; sites where most records deviate from the
    plurality (bucket sizes in the thousands); a record that is all N."""
    n, L = 1500, 2600
    codes = low_diversity(n, L, 3)
    rng = np.random.default_rng(5)
    for r in (0, 3, 700, 1499):                      # ~60 % of the sites differ: lists of ~1,500 entries
        m = rng.random(L) < 0.6
        codes[r, m] = rng.choice(CODES, size=int(m.sum()))
    codes[5, :] = 240
    codes[6, :] = 244
    for s in (0, 17, 1300, L - 1):                   # 45 % minor allele at a few sites
        m = rng.random(n) < 0.45
        codes[m, s] = 24 if codes[0, s] != 24 else 72
    eng.set_path("consensus")
    eng.upload(0, codes)
    for m in ALL:
        want = oracle.all_pairs_square(m, codes)
        got = eng.run_square(m)
        if m in da.INT_MEASURES:
            assert np.array_equal(got, want.astype(np.int64)), m
        else:
            ok = np.isclose(got, want, rtol=0, atol=1e-12, equal_nan=True)
            assert ok.all(), (m, int((~ok).sum()))
    assert eng.last_path() == "consensus"
    eng.set_path("auto")


def test_wide_tallies_for_long_alignments(eng):
    """65,536 sites or more: one 32-bit word per tally (up to four words per pair, 128 KiB of LDS)."""
    L = 70001
    a = low_diversity(40, L, 6, subs=2e-2, p_n=1e-2)
    b = low_diversity(9, L, 7, subs=2e-2, p_n=1e-2)
    a[3, :] = a[4, :]                                 # identical pair: tallies reach L
    eng.set_path("consensus")
    eng.upload(0, a)
    eng.upload(1, b)
    for m in ALL:
        om = "n_high" if m == "n" else m
        tl = eng.run_square(m, tallies=True)
        ij = oracle.pairs_square(len(a))
        for k in range(0, len(ij), 7):
            i, j = int(ij[k][0]), int(ij[k][1])
            assert list(tl[k]) == [int(x) for x in oracle.tallies(om, a[i], a[j])], (m, i, j)
        rect = eng.run_rect(m, tallies=True)          # rows: slot 0, columns: slot 1 (its own reference)
        stream = eng.run_rect(m, row_slot=1, col_slot=0, tallies=True)
        for i in (0, 3, 39):
            for j in (0, 8):
                want = [int(x) for x in oracle.tallies(om, a[i], b[j])]
                assert list(rect[i, j]) == want and list(stream[j, i]) == want, (m, i, j)
        assert eng.last_path() == "consensus"
    assert int(eng.run_square("tn93", 3, 4, tallies=True)[0][0]) > 65535
    eng.set_path("auto")


def test_rectangle_and_stream_against_another_sets_reference(eng):
    """Two files / stream mode: the rows' lists are taken against the COLUMN set's reference, also when the
    two sets have different plurality codes at many sites."""
    L = 900
    a = low_diversity(300, L, 8)
    b = low_diversity(70, L, 9)                       # another root: differs from a's at ~3/4 of the sites
    eng.set_path("consensus")
    eng.upload(0, a)
    eng.upload(1, b)
    for m in ALL:
        want = oracle.all_pairs_rect(m, a, b)
        for got in (eng.run_rect(m), eng.run_rect(m, row_slot=1, col_slot=0).T):
            if m in da.INT_MEASURES:
                assert np.array_equal(got, want.astype(np.int64)), m
            else:
                assert np.isclose(got, want, rtol=0, atol=1e-12, equal_nan=True).all(), m
    # re-uploading the column set invalidates the rows' lists (they were relative to its old reference)
    b2 = low_diversity(70, L, 10)
    eng.upload(1, b2)
    assert np.array_equal(eng.run_rect("n_high"), oracle.all_pairs_rect("n_high", a, b2).astype(np.int64))
    assert np.array_equal(eng.run_square("n_high"), oracle.all_pairs_square("n_high", a).astype(np.int64))
    eng.set_path("auto")


def test_auto_picks_by_diversity(eng):
    from tools import synth
    eng.set_path("auto")
    low = synth.alignment(7, 12000, 6000)             # SURVEY 8(d)'s diversity: ~13 differing sites per record
    eng.upload(0, low)
    d_low = eng.run_square("raw")
    assert eng.last_path() == "consensus"
    high = uniform_codes(3000, 6000, 12)              # every code equally likely: nothing to gain from lists
    eng.upload(0, high)
    eng.run_square("raw", 0, 64)
    assert eng.last_path() == "dense"
    eng.upload(0, low[:40])                           # a launch too small to pay for building the lists
    eng.run_square("raw")
    assert eng.last_path() == "dense"
    eng.upload(0, low)
    eng.set_path("dense")
    assert np.array_equal(eng.run_square("raw"), d_low, equal_nan=True)
    eng.set_path("auto")


def test_consensus_matches_fastaio(eng, golden):
    """dst_consensus against the oracle's consensus(): non-ACGT counted as A, ties to the first of A,G,C,T,
    one and two loaded sets; and the reference's own fixture (src/fastaio.rs:431-453)."""
    for seed, n, L in ((1, 1, 40), (2, 2, 300), (3, 57, 1000), (4, 2100, 333)):
        a = uniform_codes(n, L, seed) if seed % 2 else random_alignment(n, L, seed, p_gap=0.3)
        b = random_alignment(max(1, n // 2), L, seed + 100, p_ambig=0.2)
        eng.upload(0, a)
        eng.upload(1, b)
        assert np.array_equal(eng.consensus(), oracle.consensus(a)), seed
        assert np.array_equal(eng.consensus(both_slots=True), oracle.consensus(a, b)), seed
    for v in golden["consensus"]:
        recs = np.stack([oracle.encode(s.encode()) for s in v["rows"]])
        eng.upload(0, recs)
        assert eng.consensus().tolist() == v["codes"]


def test_differences_match_fastaio(eng, golden):
    """dst_differences against get_differences(): ascending sites, N / - / ? never listed."""
    for seed, n, L in ((1, 3, 15), (2, 40, 1000), (3, 9, 5000)):
        a = random_alignment(n, L, seed, p_gap=0.1, p_ambig=0.05, divergence=0.1)
        eng.upload(0, a)
        cons = oracle.consensus(a)
        for other in (cons, a[0], np.full(L, 240, np.uint8)):
            got = eng.differences(0, other)
            for r in range(n):
                assert np.array_equal(got[r].astype(np.uint64), oracle.get_differences(a[r], other)), (seed, r)
    for v in golden["get_differences"]:
        q, t = oracle.encode(v["seq"].encode()), oracle.encode(v["other"].encode())
        eng.upload(0, np.stack([q]))
        assert eng.differences(0, t)[0].tolist() == v["differences"]


def test_n_equals_the_sparse_walk_of_the_reference(eng):
    """C1's shape, complete: 100 x 10,000 `-m n` on both paths against snp_consensus() walked over the
    oracle's consensus and difference lists (src/measures.rs:28-53)."""
    from tools import synth
    codes = synth.alignment(synth.SEED ^ 1, 100, 10_000)
    cons = oracle.consensus(codes)
    diffs = [oracle.get_differences(r, cons) for r in codes]
    want = np.array([oracle.pair_distance("n", codes[int(i)], codes[int(j)], q_diffs=diffs[int(i)], t_diffs=diffs[int(j)])
                     for i, j in oracle.pairs_square(100)], np.int64)
    eng.upload(0, codes)
    for path in ("dense", "consensus", "auto"):
        eng.set_path(path)
        assert np.array_equal(eng.run_square("n"), want), path
    # the device's own precompute gives the same lists the walk used
    assert np.array_equal(eng.consensus(), cons)
    dev = eng.differences(0, cons)
    assert all(np.array_equal(dev[r].astype(np.uint64), diffs[r]) for r in range(100))
    eng.set_path("auto")


def clade_alignment(n, L, seed, clade_frac=0.35, clade_sites=0.02, subs=2e-3):
    """Phylogenetic structure: a clade (a third of the records) shares substitutions at 2 % of the sites, a
    sub-clade a few more, some columns are indel-rich — on top of sparse private differences."""
    rng = np.random.default_rng(seed)
    codes = low_diversity(n, L, seed + 1, subs=subs)
    root = codes[0].copy()
    clade = rng.random(n) < clade_frac
    sub = clade & (rng.random(n) < 0.4)
    sites = np.nonzero(rng.random(L) < clade_sites)[0]
    sub_sites = np.nonzero(rng.random(L) < clade_sites / 3)[0]
    alt = {136: 72, 72: 136, 40: 24, 24: 40}
    for s_ in sites:
        codes[clade, s_] = alt.get(int(root[s_]), 24)
    for s_ in sub_sites:
        codes[sub, s_] = 24 if root[s_] != 24 else 40
    for s_ in rng.integers(0, L, max(1, L // 200)):          # indel-rich columns: a fifth of the records gapped or N
        m = rng.random(n) < 0.2
        codes[m, s_] = rng.choice(np.array([244, 240], np.uint8), size=int(m.sum()))
    for s_ in rng.integers(0, L, max(1, L // 400)):          # ambiguity codes at a common variant
        m = rng.random(n) < 0.1
        codes[m, s_] = rng.choice(CODES[4:14], size=int(m.sum()))
    return np.ascontiguousarray(codes)


def test_hybrid_path_on_clade_structured_alignments(eng):
    """Hot columns (common variants) through the dense kernels, the rest through the lists: the same integers and
    the same device distances as either plain path, square / two files / stream order, every measure."""
    a = clade_alignment(900, 3000, 31)
    b = clade_alignment(130, 3000, 32)
    eng.upload(0, a)
    eng.upload(1, b)
    for m in ALL:
        res = {}
        for path in ("dense", "consensus", "hybrid"):
            eng.set_path(path)
            res[path] = (eng.run_square(m, tallies=True), eng.run_square(m), eng.run_rect(m, tallies=True),
                         eng.run_rect(m, row_slot=1, col_slot=0, row_begin=3, row_end=77))
            assert eng.last_path() == path, (m, path)
        for path in ("consensus", "hybrid"):
            for x, y in zip(res["dense"], res[path]):
                assert np.array_equal(x, y, equal_nan=True), (m, path)
        om = "n_high" if m == "n" else m
        ij = oracle.pairs_square(len(a))
        for k in range(0, len(ij), 997):
            i, j = int(ij[k][0]), int(ij[k][1])
            assert list(res["hybrid"][0][k]) == [int(x) for x in oracle.tallies(om, a[i], a[j])], (m, i, j)
    eng.set_path("auto")


def test_hybrid_wide_tallies_and_row_ranges(eng):
    L = 66001
    a = clade_alignment(60, L, 41, clade_sites=0.01)
    a[7] = a[8]
    eng.upload(0, a)
    for m in ("n_high", "raw", "k80", "tn93"):
        eng.set_path("dense")
        want = eng.run_square(m, tallies=True)
        eng.set_path("hybrid")
        assert np.array_equal(eng.run_square(m, tallies=True), want), m
        assert eng.last_path() == "hybrid"
        cat = np.concatenate([eng.run_square(m, rb, re, tallies=True) for rb, re in ((0, 11), (11, 12), (12, 60))])
        assert np.array_equal(cat, want), m
    eng.set_path("auto")


def test_auto_takes_the_hybrid_for_structured_data(eng):
    import torch
    n, L = 16000, 9000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    base = torch.tensor([136, 40, 72, 24], dtype=torch.uint8, device=dev)
    root = torch.randint(0, 4, (L,), device=dev, generator=g)
    idx = root.expand(n, L).clone()
    mut = torch.rand((n, L), device=dev, generator=g) < 1e-3
    idx[mut] = (idx[mut] + 1) % 4
    clade = torch.rand(n, device=dev, generator=g) < 0.33
    sites = torch.rand(L, device=dev, generator=g) < 0.02
    sel = clade[:, None] & sites[None, :]
    idx[sel] = (root.expand(n, L)[sel] + 1) % 4
    codes = base[idx].contiguous()
    pairs = n * (n - 1) // 2
    out = {p: torch.empty(pairs, dtype=torch.float64, device=dev) for p in ("auto", "dense")}
    for p in ("auto", "dense"):
        eng.set_path(p)
        eng.upload_device(0, codes.data_ptr(), n, L, L)
        eng.run_square_device("raw", 0, n, out[p].data_ptr(), pairs * 8)
        torch.cuda.synchronize()
        if p == "auto":
            assert eng.last_path() == "hybrid"
    assert torch.equal(out["auto"].view(torch.int64), out["dense"].view(torch.int64))      # every bit of every distance
    eng.set_path("auto")


def test_a_lean_set_still_serves_every_path(eng):
    """A low-diversity set uploaded while the path is not forced dense is stored lean (four base planes, the list
    lengths counted by the pack, the reference sampled from the bytes).  Whatever runs on it afterwards — the lists,
    the dense kernels (which need the derived planes built after all), tn93's base counts, the hybrid's hot columns —
    must give the same bits as a set that was packed for the dense path from the start."""
    codes = np.concatenate([low_diversity(2600, 6000, 91), clade_alignment(400, 6000, 92)])   # 0.5 n^2 L = 2.7e10
    ref = da.Engine(0)
    ref.set_path("dense")
    ref.upload(0, codes)
    want = {m: ref.run_square(m, 0, 40) for m in ALL}
    want_tallies = {m: ref.run_square(m, 5, 9, tallies=True) for m in ("raw", "k80", "tn93")}
    ref.close()
    for order in (("consensus", "dense", "hybrid"), ("dense", "hybrid", "consensus"), ("hybrid", "consensus", "dense")):
        eng.set_path("auto")
        eng.upload(0, codes)                         # lean: the path is open at upload time
        for path in order:
            eng.set_path(path)
            for m in ALL:
                got = eng.run_square(m, 0, 40)
                assert np.array_equal(got, want[m], equal_nan=True), (order, path, m)
            for m in want_tallies:
                assert np.array_equal(eng.run_square(m, 5, 9, tallies=True), want_tallies[m]), (order, path, m)
    eng.set_path("auto")


@pytest.mark.parametrize("n", [3000, 20000, 36000])
def test_every_wave_shape_of_the_fill_pass(eng, n):
    """The fill pass from the pack's slots runs as 2 x 32, 4 x 16 or 8 x 8 (records x chunks per wave) by the set's size,
    hands chunks of more than 7 differences (runs of N, gaps) to the whole wave, and leaves the marks the bucket pass
    cuts every record's list by.  Each shape, with long runs that cross chunk and 1,024-site boundaries and with runs
    inside hot columns, must give the dense path's bits — through the fused preparation and through the plain one."""
    L = 2300                                           # 18 chunks: three ranges of 1,024 sites, the last one partial
    rng = np.random.default_rng(700 + n)
    codes = low_diversity(n, L, 71 + n, subs=1.5e-3, p_n=5e-4, p_amb=1e-4)
    for r in rng.choice(n, size=60, replace=False):    # runs of N / gap of 5..900 sites anywhere in the record
        a = int(rng.integers(0, L - 5))
        codes[r, a:a + int(rng.integers(5, 900))] = 240 if rng.random() < 0.7 else 244
    codes[rng.random(n) < 0.2, 1000:1030] = 72         # hot columns straddling the 1,024 mark (hybrid path)
    codes[n // 2, :] = 240                             # one record that differs everywhere
    rows = sorted({0, 1, n // 2 - 1, n // 2, n - 2} | set(int(x) for x in rng.choice(n - 1, size=4)))
    ref = da.Engine(0)
    ref.set_path("dense")
    ref.upload(0, codes)
    want = {(m, r): ref.run_square(m, r, r + 1) for m in ("n_high", "raw", "tn93") for r in rows}
    want_t = {r: ref.run_square("tn93", r, r + 1, tallies=True) for r in rows}
    ref.close()
    # ... and the reference tallies themselves against the oracle, directly (not only through the dense path)
    for r in rows[:3]:
        for j in sorted({r + 1, n // 2, n - 1} - set(range(r + 1))):
            assert list(want_t[r][j - r - 1]) == [int(x) for x in oracle.tallies("tn93", codes[r], codes[j])], (r, j)
            assert int(want[("n_high", r)][j - r - 1]) == oracle.pair_distance("n_high", codes[r], codes[j])
    for threshold in (0.0, 1e30):                      # fused preparation / lists built at the first run
        eng.set_prep_threshold(threshold)
        for path in ("consensus", "hybrid"):
            eng.set_path("auto")
            eng.upload(0, codes)
            eng.set_path(path)
            for (m, r), w in want.items():
                assert np.array_equal(eng.run_square(m, r, r + 1), w, equal_nan=True), (threshold, path, m, r)
            for r, w in want_t.items():
                assert np.array_equal(eng.run_square("tn93", r, r + 1, tallies=True), w), (threshold, path, r)
    eng.set_prep_threshold(2e10)
    eng.set_path("auto")


@pytest.mark.parametrize("L", [3000, 70000])
def test_event_heavy_launches_on_every_path(eng, L):
    """More than one event per pair switches the pair kernel to its no-roles launch variant (all 8 waves apply a batch's
    events, then all 8 write it out; one accumulator buffer) — also under the hybrid path, whose cold sites alone can
    carry that many events, and with 32-bit accumulators (L >= 65,536).  Clade structure on top of 3 % private
    substitutions: every measure and the tallies must be the dense path's bits."""
    n = 1500 if L < 65536 else 500
    codes = clade_alignment(n, L, 311 + L, subs=0.03)
    ref = da.Engine(0)
    ref.set_path("dense")
    ref.upload(0, codes)
    rows = (0, n // 3, n - 2)
    want = {(m, r): ref.run_square(m, r, r + 2 if r + 2 < n else r + 1) for m in ALL for r in rows}
    want_t = {m: ref.run_square(m, 3, 6, tallies=True) for m in ("raw", "k80", "tn93")}
    ref.close()
    # the no-roles variant (EW == 8) against the oracle directly: rows 3..5 of the tallies, a handful of columns
    for m in ("raw", "k80", "tn93"):
        for j in (4, 7, n // 2, n - 1):
            at = j - 4                                   # pair (3, j) is entry j - 4 of row 3's slab
            assert list(want_t[m][at]) == [int(x) for x in oracle.tallies(m, codes[3], codes[j])], (m, j)
    for path in ("consensus", "hybrid"):
        eng.set_path("auto")
        eng.upload(0, codes)
        eng.set_path(path)
        for (m, r), w in want.items():
            assert np.array_equal(eng.run_square(m, r, r + 2 if r + 2 < n else r + 1), w, equal_nan=True), (path, m, r)
        for m, w in want_t.items():
            got = eng.run_square(m, 3, 6, tallies=True)
            assert np.array_equal(got, w), (path, m)
            for j in (4, n // 2, n - 1):                 # and this path's own tallies against the oracle
                assert list(got[j - 4]) == [int(x) for x in oracle.tallies(m, codes[3], codes[j])], (path, m, j)
        for m in ("jc69", "tn93"):                       # device distances of the variant: BASELINE's 1e-12
            d = eng.run_square(m, 3, 4)
            for j in (4, n // 2, n - 1):
                w = oracle.pair_distance(m, codes[3], codes[j])
                assert (np.isnan(w) and np.isnan(d[j - 4])) or d[j - 4] == w or abs(d[j - 4] - w) <= 1e-12, (path, m, j)
    eng.set_path("auto")
