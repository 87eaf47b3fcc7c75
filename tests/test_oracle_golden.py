"""CPU: the oracle against every known-answer vector the reference's own tests hold
(tests/golden/reference_vectors.json), plus an independent per-site restatement in pure Python
over the exhaustive 17x17 code-pair table."""
import itertools
import math

import numpy as np
import pytest

import oracle
from helpers import CODES, KNOWN, random_alignment, uniform_codes


def test_encoding_table_matches_reference_listing():
    # src/encoding.rs:7-38, letter by letter
    want = dict(A=136, G=72, C=40, T=24, R=192, M=160, W=144, S=96, K=80, Y=48, V=224, H=176,
                D=208, B=112, N=240)
    t = oracle.encoding_array()
    for ch, v in want.items():
        assert t[ord(ch)] == v and t[ord(ch.lower())] == v
    assert t[ord("-")] == 244 and t[ord("?")] == 242
    mapped = {ord(c) for c in want} | {ord(c.lower()) for c in want} | {ord("-"), ord("?")}
    assert all(t[b] == 0 for b in range(256) if b not in mapped)
    assert sorted(set(int(x) for x in t if x)) == sorted(int(c) for c in CODES)


def test_encode_golden(golden):
    for v in golden["encode"]:
        assert oracle.encode(v["seq"].encode()).tolist() == v["codes"]
    with pytest.raises(ValueError):
        oracle.encode(b"ATGU")  # 'U' is unmapped -> invalid (encoding.rs: absent)
    with pytest.raises(ValueError):
        oracle.encode(b"AT.G")


def test_count_bases_golden(golden):
    for v in golden["count_bases"]:
        a, t, g, c = oracle.count_bases(oracle.encode(v["seq"].encode()))
        assert (a, t, g, c) == (v["A"], v["T"], v["G"], v["C"])
        _, counts = oracle.encode_count_bases(v["seq"].encode())
        assert counts.tolist() == [v["A"], v["T"], v["G"], v["C"]]
    # fastaio.rs:136-142 counts raw upper-case characters only; count_bases counts by code
    codes, counts = oracle.encode_count_bases(b"aAtTgGcC")
    assert counts.tolist() == [1, 1, 1, 1]
    assert oracle.count_bases(codes).tolist() == [2, 2, 2, 2]


def test_get_differences_golden(golden):
    for v in golden["get_differences"]:
        d = oracle.get_differences(oracle.encode(v["seq"].encode()), oracle.encode(v["other"].encode()))
        assert d.tolist() == v["differences"]
    # N / - / ? never enter the list (seq[i] < 240)
    assert oracle.get_differences(oracle.encode(b"N-?R"), oracle.encode(b"AAAA")).tolist() == [3]


def test_consensus_golden(golden):
    for v in golden["consensus"]:
        rows = np.stack([oracle.encode(r.encode()) for r in v["rows"]])
        assert oracle.consensus(rows).tolist() == v["codes"]
    # non-ACGT codes are counted in the A bucket (fastaio.rs:295-302)
    rows = np.stack([oracle.encode(b"N"), oracle.encode(b"R"), oracle.encode(b"T")])
    assert oracle.consensus(rows).tolist() == [136]


def test_measures_golden(golden):
    for v in golden["measures"]:
        q = oracle.encode(v["query"].encode())
        t = oracle.encode(v["target"].encode())
        if v["measure"] == "n":
            cons = oracle.consensus(np.stack([q, t]))
            got = oracle.pair_distance("n", q, t, q_diffs=oracle.get_differences(q, cons),
                                       t_diffs=oracle.get_differences(t, cons))
        else:
            got = oracle.pair_distance(v["measure"], q, t)
        if "int" in v:
            assert got == v["int"], v
        else:
            assert got == float.fromhex(v["hex"]), v  # exact, like the reference's assert_eq!


def test_pair_order_golden(golden):
    for v in golden["pairs_square"]:
        assert oracle.pairs_square(v["n"]).tolist() == v["pairs"]
    for v in golden["pairs_rectangle"]:
        assert oracle.pairs_rectangle(v["n1"], v["n2"]).tolist() == v["pairs"]


def test_tsv_golden(golden):
    for v in golden["tsv"]:
        ids1 = [r[0] for r in v["loaded"]]
        a = np.stack([oracle.encode(r[1].encode()) for r in v["loaded"]])
        if v["mode"] == "square":
            d = oracle.all_pairs_square(v["measure"], a)
            ij = oracle.pairs_square(len(a))
            text = oracle.tsv(ids1, ids1, ij, [int(x) for x in d])
        else:
            other = v.get("streamed") or v["second"]
            ids2 = [r[0] for r in other]
            b = np.stack([oracle.encode(r[1].encode()) for r in other])
            d = oracle.all_pairs_rect(v["measure"], a, b)
            if v["mode"] == "stream":  # streamed record outer, loaded inner (lib.rs:323-331)
                ij = [(i, j) for j in range(len(b)) for i in range(len(a))]
            else:
                ij = [(i, j) for i in range(len(a)) for j in range(len(b))]
            text = oracle.tsv(ids1, ids2, ij, [int(d[i, j]) for i, j in ij])
        assert text == v["expected"], v


def test_float_formatting_rust_display():
    # Rust {:.12}: lib.rs:631
    f = oracle.format_distance
    assert f(2.0 / 15.0) == "0.133333333333"
    assert f(float("nan")) == "NaN"
    assert f(float("inf")) == "inf"
    assert f(float("-inf")) == "-inf"
    assert f(-0.0) == "-0.000000000000"
    assert f(0.0) == "0.000000000000"
    assert f(7) == "7"


# ---------------------------------------------------------------- independent restatement --
def _site(q, t):
    """Per-site contribution, restated straight from measures.rs in pure Python (no oracle)."""
    same = (q & 8) == 8 and q == t
    diff = (not same) and (q & t) < 16
    pur = lambda x: (x & 55) == 0
    pyr = lambda x: (x & 199) == 0
    r = dict(snp=int((q & t) < 16), raw_n=int(diff), raw_d=int(same or diff))
    k_l = k_ts = k_tv = 0
    if same:
        k_l = 1
    elif diff:
        if (pur(q) and pur(t)) or (pyr(q) and pyr(t)):
            k_ts = k_l = 1
        elif (pur(q) and pyr(t)) or (pyr(q) and pur(t)):
            k_tv = k_l = 1
    r.update(k_l=k_l, k_ts=k_ts, k_tv=k_tv)
    t_l = t_d = t_p1 = t_p2 = 0
    if same:
        t_l = 1
    elif (q & t) < 16 and (q & 8) == 8 and (t & 8) == 8:
        t_d = t_l = 1
        if (q | t) == 200:
            t_p1 = 1
        elif (q | t) == 56:
            t_p2 = 1
    r.update(t_l=t_l, t_d=t_d, t_p1=t_p1, t_p2=t_p2)
    return r


def test_truth_table_17x17():
    """Every ordered pair of valid codes as a 1-site alignment, oracle vs the restatement."""
    dropped = 0
    for q, t in itertools.product(CODES.tolist(), repeat=2):
        s = _site(q, t)
        qa, ta = np.array([q], np.uint8), np.array([t], np.uint8)
        assert oracle.tallies("n_high", qa, ta).tolist() == [s["snp"]]
        assert oracle.tallies("raw", qa, ta).tolist() == [s["raw_n"], s["raw_d"]]
        assert oracle.tallies("k80", qa, ta).tolist() == [s["k_l"], s["k_ts"], s["k_tv"]]
        assert oracle.tallies("tn93", qa, ta).tolist() == [s["t_l"], s["t_d"], s["t_p1"], s["t_p2"]]
        # only the high nibble decides: certainly-different <=> no shared base bit
        assert s["snp"] == int(((q >> 4) & (t >> 4)) == 0)
        dropped += int(s["raw_n"] == 1 and s["k_l"] == 0)
    assert dropped == 28  # the 28 ordered "different" code pairs k80 drops


def test_n_equals_n_high_on_random_alignments():
    for seed, gen in ((1, random_alignment), (2, uniform_codes)):
        codes = gen(12, 257, seed)
        a = oracle.all_pairs_square("n", codes)
        b = oracle.all_pairs_square("n_high", codes)
        assert np.array_equal(a, b)


def test_all_pairs_matches_per_pair_and_threads():
    codes = random_alignment(9, 301, 5)
    ij = oracle.pairs_square(9)
    for m in ("n_high", "raw", "jc69", "k80", "tn93"):
        one = oracle.all_pairs_square(m, codes, threads=1)
        three = oracle.all_pairs_square(m, codes, threads=3)
        assert np.array_equal(one, three, equal_nan=True)
        for p, (i, j) in enumerate(ij):
            want = oracle.pair_distance(m, codes[i], codes[j])
            assert (one[p] == want) or (math.isnan(one[p]) and math.isnan(want))
    sub = oracle.all_pairs_square("raw", codes, pair_range=(7, 20), threads=2)
    assert np.array_equal(sub, oracle.all_pairs_square("raw", codes)[7:20], equal_nan=True)


def test_tallies_then_finalize_equals_direct():
    codes = random_alignment(6, 500, 11)
    counts = oracle.count_bases_matrix(codes)
    for m in ("raw", "jc69", "k80", "tn93"):
        for i, j in oracle.pairs_square(6):
            i, j = int(i), int(j)
            tl = oracle.tallies(m, codes[i], codes[j])
            got = oracle.finalize(m, tl, counts[i], counts[j])
            want = oracle.pair_distance(m, codes[i], codes[j])
            assert got == want or (math.isnan(got) and math.isnan(want))


def test_edge_values():
    a = oracle.encode(b"ACGT")
    n = oracle.encode(b"NNNN")
    assert math.isnan(oracle.pair_distance("raw", a, n))          # 0/0
    assert oracle.pair_distance("raw", a, a) == 0.0
    jc = oracle.pair_distance("jc69", a, a)
    assert jc == 0.0 and math.copysign(1.0, jc) == -1.0            # -0.75*ln(1) = -0.0
    k = oracle.pair_distance("k80", a, a)
    assert k == 0.0 and math.copysign(1.0, k) == -1.0
    tn = oracle.pair_distance("tn93", a, a)
    assert tn == 0.0 and math.copysign(1.0, tn) == 1.0             # measures.rs:188-190
    far = oracle.encode(b"CATG")
    assert oracle.pair_distance("jc69", a, far) != oracle.pair_distance("jc69", a, far)  # p=1 -> NaN
    assert oracle.pair_distance("n_high", a, far) == 4
